#!/bin/bash
# round 2, call W: what the default mode (ESL + ERT) spends: lit against unlit, TRILINEAR and NEAREST
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for s in trilinear nearest; do for l in 0.6 0; do echo "== default mode $s light $l"; timeout -k 10 300 python scripts/perf_probe.py --mode default --sampling $s --light $l || exit 1; done; done
