#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03l; mkdir -p $O
for block in 1 0; do for light in 0.6 0; do
timeout -k 10 200 python scripts/perf_probe.py --mode nooptims --views 0 --block $block --light $light > $O/p.json 2>$O/probe.err || { tail -5 $O/probe.err; exit 1; }
python -c 'import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], sys.argv[3], d["mean_ms"])' $O/p.json block=$block light=$light
done; done
bash scripts/gpu_pmc.sh $O/pmc sq1,sq2,tcc --mode nooptims --views 0 || exit 1
python scripts/pmc_per_view.py $O/pmc 4 blockmarch
python scripts/pmc_per_view.py $O/pmc 4 raymarch
