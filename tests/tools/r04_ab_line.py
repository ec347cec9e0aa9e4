import json,sys
o=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); e=o["extra"]
print(o["value"],o["ms_per_step"],e["msm_phase_ms"]["msm_sort"],e["msm_phase_ms"]["msm_accumulate"],{k:v["pipelined_ms"] for k,v in e["msm_sweep"]["sizes"].items()}, e["config5"]["msm_ms"])
