#!/bin/bash
# rocprofv3 kernel stats of the DiT-L/2 batch-64 forward with LFM_OPT_FOLD_LN on (default) and off.  usage: tools/stats_fold.sh <tag>
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
cat > /tmp/fwd.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
from lfm_amd import hip
from lfm_amd.models import DiT_models
dev = torch.device("cuda:0")
m = DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0)
for p in m.parameters():
    if not bool(p.any()): torch.nn.init.normal_(p, std=0.02)
m = m.to(dev).eval()
x = torch.randn(64, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev)
hip.set_option(hip.OPT_FOLD_LN, int(sys.argv[1]))
for _ in range(8): m(t, x)
torch.cuda.synchronize()
PY
for v in 1 0; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fold$v -o f -- python /tmp/fwd.py $v > $O/fold$v.log 2>&1
  cp $(find $O/fold$v -name "*kernel_stats.csv" | head -1) $O/fold${v}_kernel_stats.csv
  echo "== fold=$v"; head -16 $O/fold${v}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
done
