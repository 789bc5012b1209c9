#!/bin/bash
# kernel stats of a few DiT-L/2 batch-64 forwards (auto kernel selection)
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/statsprobe; rm -rf $O; mkdir -p $O
cat > /tmp/fw.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
from lfm_amd.models import DiT_models
dev = torch.device("cuda:0")
m = DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0).to(dev).eval()
x = torch.randn(64, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev)
for _ in range(12): m(t, x)
torch.cuda.synchronize()
PY
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python /tmp/fw.py > $O/log.txt 2>&1
cd $R && python - <<'PY'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/statsprobe/s_kernel_stats.csv')))
for r in rows[:9]:
    print(r['Name'][:84].ljust(84), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:9.1f} us", r['Percentage'])
PY
