"""Config 5 through pipeline.PipelineLanes: the block range cut into GROUPS that LANES (a thread + rcx_ctx + HIP stream each)
work through side by side.  Prints encode / decode seconds per (lanes, groups) (best of 3 after a warm-up) and checks the
coded bytes against the one-lane pipeline's.
usage: python benchmarks/r4_pipe_lanes.py [scale] [lanes:groups,lanes:groups,...]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])


def main():
    import torch
    import rust_compress_amd as R
    from rust_compress_amd import synth, pipeline as P
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    combos = [tuple(int(v) for v in x.split(":")) for x in (sys.argv[2] if len(sys.argv) > 2 else "1:1,2:2,2:4,2:6,3:3,3:6").split(",")]
    dev = torch.device("cuda", 0)
    BLOCK = 262144
    total = int(1e9 * scale)
    lens = np.array([BLOCK] * (total // BLOCK) + ([total % BLOCK] if total % BLOCK else []), dtype=np.int64)
    data = np.concatenate([synth.gen("text", min(BLOCK * 256, total - s), 0xC0 + s) for s in range(0, total, BLOCK * 256)])[:total]
    raw = torch.from_numpy(data).to(dev)
    ctx = R.Context(0)
    ref = P.BwtDcAri(ctx, dev)
    comp0, coff0, clen0, praw0, _ = ref.encode(raw, lens)
    torch.cuda.synchronize()
    pay0 = [comp0[int(o):int(o) + int(n)].cpu().numpy().tobytes() for o, n in list(zip(coff0.reshape(-1), clen0.reshape(-1)))[:: max(1, clen0.size // 512)]]
    for L, G in combos:
        pipe = P.PipelineLanes(dev, lanes=L, groups=G)
        te = td = 1e9
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            comp, coff, clen, praw, _ = pipe.encode(raw, lens)
            torch.cuda.synchronize(); e_ = time.perf_counter() - t0
            t0 = time.perf_counter()
            back = pipe.decode(comp, coff, clen, praw, lens)
            torch.cuda.synchronize(); d_ = time.perf_counter() - t0
            if rep:
                te, td = min(te, e_), min(td, d_)
        pay = [comp[int(o):int(o) + int(n)].cpu().numpy().tobytes() for o, n in list(zip(coff.reshape(-1), clen.reshape(-1)))[:: max(1, clen.size // 512)]]
        same = bool(np.array_equal(clen, clen0)) and bool(np.array_equal(praw, praw0)) and pay == pay0
        print(json.dumps({"lanes": L, "groups": G, "encode_s": round(te, 4), "decode_s": round(td, 4), "roundtrip": bool(torch.equal(back, raw)),
                          "coded_bytes_equal_one_lane": same}), flush=True)
        pipe.close()
        del pipe, comp, back
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
