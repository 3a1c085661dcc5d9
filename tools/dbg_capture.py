"""Reproducer: a stream forked inside hipGraph capture that forks again crashes hipStreamEndCapture
(ROCm 7.0 runtime bundled with torch 2.10); sibling forks from the capture stream are fine.  This is
why EnsembleTBPTTStep composes separately captured member graphs as child-graph nodes instead."""
import subprocess
import sys

CASES = ["sibling", "nested"]

if len(sys.argv) == 1:
    for c in CASES:
        r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True)
        print(c, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1], flush=True)
    sys.exit(0)

import torch

case = sys.argv[1]
dev = torch.device("cuda", 0)
st, side = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
x = torch.ones(1024, device=dev)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    cur = torch.cuda.current_stream(dev)
    y0 = x * 2
    if case == "sibling":
        st.wait_stream(cur)
        side.wait_stream(cur)
        with torch.cuda.stream(st):
            a = x * 3
        with torch.cuda.stream(side):
            b = x * 4
        cur.wait_stream(st)
        cur.wait_stream(side)
    else:
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            a = x * 3
            side.wait_stream(st)
            with torch.cuda.stream(side):
                b = a * 4
            c = a * 5
            st.wait_stream(side)
            d = b + c
        cur.wait_stream(st)
    z = y0 + 1
graph.replay()
torch.cuda.synchronize()
print("ok")
