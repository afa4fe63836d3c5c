#!/bin/bash
# round-2 GPU job E: parity suite on the fused split-K / bn-fold / dual-quantizer build, A/B of the fused reduction
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02e_tests.log 2>&1; echo "tests rc=$?" | tee gpurun_out/r02e_status.txt
tail -4 gpurun_out/r02e_tests.log
python bench.py --no-cpu-baseline --no-others --steps 20 > gpurun_out/r02e_bench_fused.json 2> gpurun_out/r02e_bench_fused.err; echo "bench fused rc=$?" | tee -a gpurun_out/r02e_status.txt
RG_SPLITK_FUSED=0 python bench.py --no-cpu-baseline --no-others --steps 20 > gpurun_out/r02e_bench_unfused.json 2> gpurun_out/r02e_bench_unfused.err; echo "bench unfused rc=$?" | tee -a gpurun_out/r02e_status.txt
python bench.py --config 5 --no-cpu-baseline --no-others --steps 20 > gpurun_out/r02e_bench5.json 2> gpurun_out/r02e_bench5.err; echo "bench5 rc=$?" | tee -a gpurun_out/r02e_status.txt
python bench.py --config 3 --no-cpu-baseline --no-others --steps 20 > gpurun_out/r02e_bench3.json 2> gpurun_out/r02e_bench3.err; echo "bench3 rc=$?" | tee -a gpurun_out/r02e_status.txt
python - <<'PY'
import json
for n in ("fused","unfused","5","3"):
    f = "gpurun_out/r02e_bench_%s.json" % n if n in ("fused","unfused") else "gpurun_out/r02e_bench%s.json" % n
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(n, d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"], r["achieved"], r["frac"], r["kernel_ms_per_step"], r["launches_per_step"],
              {k: (v["ms_per_step"], v["launches"]) for k, v in r["by_family"].items()})
    except Exception as e:
        print(n, "ERR", e)
PY
