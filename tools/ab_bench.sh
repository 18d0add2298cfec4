#!/bin/bash
# A/B runs of bench.py on one box, alternating processes:  tools/ab_bench.sh TAG ROUNDS "ENV1" "ENV2" ... [-- bench args]
# every ENVk is a (possibly empty) string of VAR=value assignments; prints clips/s per run.
TAG=$1; ROUNDS=$2; shift 2
ENVS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
[ "$1" = "--" ] && shift
cd $GRAFT_REPO_ROOT
for r in $(seq 1 $ROUNDS); do
  for i in "${!ENVS[@]}"; do
    out=gpurun_out/ab_${TAG}_${i}_${r}.json
    env ${ENVS[$i]} timeout -k 10 300 python bench.py --no-cpu-baseline --no-kernel-rooflines --no-other-workloads --steps 4 "$@" > $out 2> gpurun_out/ab_${TAG}_${i}_${r}.err
    python - "$out" "${ENVS[$i]}" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(f"[{sys.argv[2] or 'default'}] {d['value']:.2f} clips/s  model {d['roofline_model']['avg_ms'] / d['config']['optimizer_iterations_per_step']:.4f} ms/iteration", flush=True)
except Exception as exc:
    print(f"[{sys.argv[2]}] failed: {exc}", flush=True)
PY
  done
done
