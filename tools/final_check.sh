#!/bin/bash
# what the driver does at round end, once more on a fresh box: smoke(), then the bench line in the driver's form
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/bench_driver_form.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_driver_form.json").read().strip().splitlines()[-1])
print(d["metric"], round(d["value"]), d["unit"], "ms_per_step", round(d["ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 3), "cpu", d["cpu_baseline"]["value"],
      "config5", round(d["other_configs"]["config5_N32768_K200_32bit"]["sweeps_per_s"]), "moving", round(d["moving_regime"]["sweeps_per_s"]), round(d["moving_regime"]["sweeps_per_s_incremental_mode"]))
PY
