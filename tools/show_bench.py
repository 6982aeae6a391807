"""One line per bench.py JSON line.  Usage: python tools/show_bench.py FILE [FILE ...]   (or: ... | python tools/show_bench.py -)
Reads files named on the command line; standard input only when asked for with "-" (a run on the GPU box that waits for
input that never comes is killed for its silence after seven minutes)."""
import fileinput,json,sys
if len(sys.argv) < 2:
    sys.exit(__doc__)
for line in fileinput.input(sys.argv[1:]):
    line=line.strip()
    if not line.startswith("{"): continue
    d=json.loads(line)
    print("value %.1f  ms/step %.4f  kernel %.4f  pack %.4f  gpu_side %.4f  depth %s streams %s  valu_frac %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["pack_kernel_ms"], d["gpu_side_ms_per_step"], d["config"].get("batches_in_flight"), d["config"].get("compute_streams"), d["roofline"].get("valu",{}).get("frac")))
