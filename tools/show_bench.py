import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith("{"): continue
    d=json.loads(line)
    print("value %.1f  ms/step %.4f  kernel %.4f  pack %.4f  gpu_side %.4f  depth %s streams %s  valu_frac %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["pack_kernel_ms"], d["gpu_side_ms_per_step"], d["config"].get("batches_in_flight"), d["config"].get("compute_streams"), d["roofline"].get("valu",{}).get("frac")))
