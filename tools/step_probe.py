import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, ribbit_amd
from ribbit_amd.simulate import simulate_sequence
seq,_ = simulate_sequence(100_000_000, 2, 2, 100)
d = torch.frombuffer(bytearray(seq), dtype=torch.uint8).cuda()
depth=3
scs=[ribbit_amd.Scanner(2,100) for _ in range(depth)]
st=torch.cuda.Stream()
for h in scs: h.set_stream(st.cuda_stream)
def issue(k):
    h=scs[k%depth]; h.load_record_device(d.data_ptr(), d.numel()); h.scan_perfect_begin()
def run(n):
    issued=0
    for k in range(1, min(depth-1,n)+1): issued+=1; issue(issued)
    for k in range(1,n+1):
        p = scs[k%depth].scan_perfect_end(wait=False) if issued>=k else None
        if issued<n: issued+=1; issue(issued)
        if p is None: p = scs[k%depth].scan_perfect_end(wait=False)
        scs[k%depth].scan_perfect_wait()
run(5); torch.cuda.synchronize()
t=time.perf_counter(); run(40); torch.cuda.synchronize(); dt=time.perf_counter()-t
print("ms/step %.4f" % (dt/40*1e3))
