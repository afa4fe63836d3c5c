#!/bin/bash
# round-2 GPU job F: hipGraph capture tests, then eager vs graph A/B on the same box for every configuration
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_graph_gpu.py tests/test_f8_gpu.py -x -q -m gpu > gpurun_out/r02f_graph_tests.log 2>&1; echo "graph tests rc=$?" | tee gpurun_out/r02f_status.txt
tail -15 gpurun_out/r02f_graph_tests.log | cut -c1-220
for c in 2 3 4a 4b 5; do
  for g in off on; do
    timeout -k 10 300 python bench.py --config $c --graph $g --no-cpu-baseline --no-others --steps 20 --profile-steps 0 > gpurun_out/r02f_bench_${c}_${g}.json 2> gpurun_out/r02f_bench_${c}_${g}.err
    echo "bench $c graph=$g rc=$?" | tee -a gpurun_out/r02f_status.txt
  done
done
python - <<'PY'
import json
for c in ("2","3","4a","4b","5"):
    for g in ("off","on"):
        try:
            d = json.loads(open("gpurun_out/r02f_bench_%s_%s.json" % (c, g)).read().strip().splitlines()[-1])
            print(c, g, d["value"], d["ms_per_step"], "host", d["host_enqueue_ms_per_step"], d["launch"][:20], {k: round(v, 4) for k, v in list(d["losses"].items())[:3]})
        except Exception as e:
            print(c, g, "ERR", e)
PY
