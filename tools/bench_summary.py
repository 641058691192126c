"""Short view of a bench.py JSON line: python tools/bench_summary.py gpurun_out/x.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms", d["ms_per_step"], "value %.3g" % d["value"], "test", d["test"]["value"], d["test"]["ms_per_batch"],
      d["test"]["single_sample_latency_ms"])
print("stages", d["stages_ms"])
r = d["roofline"]
print("roof", r["kernel"][:30], r["frac"], r.get("frac_algorithmic_vs_fp32_mfma"), r.get("other_tile_modes"))
o = d["roofline_other"]
print("other", o["bound"], o["frac"], o.get("k_pick_ms"), o.get("k_rescore_ms"))
print("lat", {k: v for k, v in d["test"]["latency"].items() if k in ("ms_per_call", "launch_floor_us", "kernel_us_sum")})
print("test roof", d["test"]["roofline"]["stage_ms_per_batch"], d["test"]["roofline"]["fp64_valu_frac"],
      d["test"]["roofline"]["zscore_gather"]["l2_frac"])
e = d.get("extra") or {}
print("extra", e.get("ms_per_step"), e.get("stages_ms"), e.get("k_gram_frac_of_mfma_peak"),
      e.get("k_gram_ms_other_tile_modes"), e.get("error"))
print("extra cpu", e.get("cpu_baseline"))
t = e.get("test_50kb") or {}
print("t50", {k: v for k, v in t.items() if k not in ("roofline",)})
print("t50 roof", (t.get("roofline") or {}).get("stage_ms_per_batch"), (t.get("roofline") or {}).get("fp64_valu_frac"),
      (t.get("roofline") or {}).get("zscore_gather"))
c = d.get("cpu_baseline") or {}
print("cpu", {k: v for k, v in c.items() if k != "port_vs_reference"})
print("prep", d["prep"])
