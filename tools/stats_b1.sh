#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/statsb1; rm -rf $O; mkdir -p $O
cat > /tmp/fw1.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
from lfm_amd.models import DiT_models
dev = torch.device("cuda:0")
m = DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0).to(dev).eval()
x = torch.randn(1, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev)
for _ in range(20): m(t, x)
torch.cuda.synchronize()
PY
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python /tmp/fw1.py > $O/log.txt 2>&1
cd $R && python - <<'PY'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/statsb1/s_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("sum of kernel time per forward (us):", tot/20/1e3)
for r in rows[:12]:
    print(r['Name'][:84].ljust(84), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:9.1f} us", r['Percentage'])
PY
