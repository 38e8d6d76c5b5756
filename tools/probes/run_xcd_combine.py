"""tools/probes/run_xcd_combine.py [out.json] -- see xcd_combine.hip: microseconds per launch of 256 workgroups (one per CU) that combine groups of G partial tiles inside the
launch -- partners on ONE XCD with workgroup-scope (sc0) accesses, partners on one XCD with agent-scope accesses, partners on different XCDs (agent scope) -- against the same
grid without any exchange; the difference is what a combine costs.  Every variant's result is checked (the sum of the G synthetic partials), and the XCC id every block ran on
is read back: the "block b runs on XCD b % 8" assumption is measured."""
import ctypes as C
import json
import subprocess
import sys
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
so, src = HERE / "libxcdcombine.so", HERE / "xcd_combine.hip"
if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
    subprocess.run(["hipcc", "-O3", "-std=c++20", "-shared", "-fPIC", "--offload-arch=gfx950", str(src), "-o", str(so)], check=True)
lib = C.CDLL(str(so))
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
blocks = 256
out = {"blocks": blocks, "runs": []}
xcc = torch.full((blocks,), 99, dtype=torch.int32, device=dev)


def time_variant(scope, per, G, partners, baseline, launches=200, reps=7):
    tile_f4 = 256 * per
    slabs = torch.zeros(blocks * tile_f4 * 4, dtype=torch.float32, device=dev)
    tickets = torch.zeros(blocks, dtype=torch.int32, device=dev)
    res = torch.zeros(blocks * tile_f4 * 2, dtype=torch.int32, device=dev)

    def launch():
        rc = lib.xc_launch(scope, per, C.c_void_p(slabs.data_ptr()), C.c_void_p(tickets.data_ptr()), C.c_void_p(res.data_ptr()), C.c_void_p(xcc.data_ptr()), blocks, G, partners,
                           baseline, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    with torch.cuda.stream(stream):
        launch()
        stream.synchronize()
        if not baseline:   # check: tile t = sum over ranks of (x + rank), x = ((tid * per + i) % 97) / 4 + lane offsets 0..3
            got = res.view(torch.bfloat16).float().view(-1, tile_f4, 4)[: blocks // G]
            idx = (torch.arange(256, device=dev)[None, :] * per + torch.arange(per, device=dev)[:, None]) % 97      # [per, 256] -> float4 index i * 256 + tid
            x = idx.float().reshape(-1) * 0.25
            want = (G * x[:, None] + sum(range(G)) + G * torch.arange(4, device=dev)[None, :].float())
            want = want.bfloat16().float()
            ok = torch.allclose(got, want[None].expand_as(got), rtol=1e-2, atol=1e-2)
            assert ok, (scope, per, G, partners, (got - want[None]).abs().max().item())
            assert int(tickets.abs().sum().item()) == 0
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(launches):
                launch()
        for _ in range(3):
            g.replay()
        stream.synchronize()
        us = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            g.replay()
            e1.record(stream)
            stream.synchronize()
            us.append(e0.elapsed_time(e1) * 1e3 / launches)
    return sorted(us)[len(us) // 2]


for per, label in ((4, "16 x 256 f32 (16 KB)"), (16, "64 x 256 f32 (64 KB)"), (32, "128 x 256 f32 (128 KB)")):
    base = time_variant(1, per, 1, 1, 1)
    out["runs"].append({"tile": label, "variant": "no exchange (every workgroup writes its tile)", "us": round(base, 2)})
    print(json.dumps(out["runs"][-1]), flush=True)
    for G in (2, 4):
        for scope, partners, name in ((0, 0, "same XCD, workgroup scope (sc0: stays in that L2)"), (1, 0, "same XCD, agent scope (sc1)"), (1, 1, "neighbouring blocks = different XCDs, agent scope (sc1)")):
            us = time_variant(scope, per, G, partners, 0)
            out["runs"].append({"tile": label, "G": G, "variant": name, "us": round(us, 2), "combine_us": round(us - base, 2)})
            print(json.dumps(out["runs"][-1]), flush=True)
ids = xcc.cpu().tolist()
out["xcc_of_block_first_32"] = ids[:32]
out["block_b_runs_on_xcd_b_mod_8"] = all(ids[b] == ids[b % 8] for b in range(blocks)) and len(set(ids[:8])) == 8
print("XCC ids of blocks 0-15:", ids[:16], "-> b % 8 rule holds:", out["block_b_runs_on_xcd_b_mod_8"])
if len(sys.argv) > 1:
    Path(sys.argv[1]).write_text(json.dumps(out, indent=1))
