"""Reflow a Markdown file to a column limit: paragraphs and list items are re-wrapped, fenced code is left alone, tables with a row wider than the limit
become bullet lists (first cell bold, the other cells labelled by their column headers).  usage: python tools/md_reflow.py FILE [width]"""
import re, sys, textwrap
path = sys.argv[1]; W = int(sys.argv[2]) if len(sys.argv) > 2 else 118
lines = open(path).read().split("\n")
out, i = [], 0
def cells(row): return [c.strip() for c in row.strip().strip("|").split("|")]
def wrap(text, first, rest):
    return textwrap.wrap(text, W, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False) or [first.rstrip()]
while i < len(lines):
    ln = lines[i]
    if ln.startswith("```"):
        out.append(ln); i += 1
        while i < len(lines) and not lines[i].startswith("```"): out.append(lines[i]); i += 1
        if i < len(lines): out.append(lines[i]); i += 1
        continue
    if ln.startswith("|") and i + 1 < len(lines) and re.match(r"^\|[\s:|-]+\|$", lines[i + 1]):
        j = i
        while j < len(lines) and lines[j].startswith("|"): j += 1
        block = lines[i:j]
        if max(len(b) for b in block) <= W + 2:
            out.extend(block)
        else:
            hdr = cells(block[0])
            for row in block[2:]:
                c = cells(row)
                head = f"* **{c[0]}**" if not c[0].startswith("**") else f"* {c[0]}"
                parts = []
                for h, v in zip(hdr[1:], c[1:]):
                    if v: parts.append(f"{h}: {v}" if len(hdr) > 2 else v)
                out.extend(wrap(head + " -- " + ";  ".join(parts), "", "  "))
        i = j
        continue
    m = re.match(r"^(\s*)([*-]|\d+\.)\s+", ln)
    if m:  # list item: gather continuation lines (indented, not a new item)
        ind = m.group(1); bullet = ln[: m.end()]; text = ln[m.end():]
        i += 1
        while i < len(lines) and lines[i].strip() and not re.match(r"^\s*([*-]|\d+\.)\s+", lines[i]) and not lines[i].startswith(("#", "|", "```")) and lines[i].startswith(" "):
            text += " " + lines[i].strip(); i += 1
        out.extend(wrap(text, bullet, ind + "  "))
        continue
    if ln.strip() and not ln.startswith(("#", "|", ">")):
        text = ln.strip(); i += 1
        while i < len(lines) and lines[i].strip() and not lines[i].startswith(("#", "|", "```", ">")) and not re.match(r"^\s*([*-]|\d+\.)\s+", lines[i]):
            text += " " + lines[i].strip(); i += 1
        out.extend(wrap(text, "", ""))
        continue
    out.append(ln); i += 1
open(path, "w").write("\n".join(out))
print(path, len(out), "lines, longest", max(len(l) for l in out))
