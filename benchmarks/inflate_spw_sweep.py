import sys, os, json, time, zlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from multiprocessing import Pool
import rust_compress_amd as R
from rust_compress_amd import _native as N, synth, batch as B
def _z(a): return zlib.compress(a[1], (1,6,9)[a[0]%3])
if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    ctx = R.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for nb, BLOCK in ((65536, 16384), (32768, 16384), (16384, 16384), (4096, 16384), (1024, 16384), (1024, 262144), (128, 1 << 20)):
        raw_np = synth.gen_blocks("text", nb, BLOCK, 0x5A11)
        with Pool(32) as pool:
            members = pool.map(_z, [(i, raw_np[i*BLOCK:(i+1)*BLOCK].tobytes()) for i in range(nb)], chunksize=512)
        base, off, lens = B.pack(members)
        ar = np.arange(nb, dtype=np.int64)
        db = R.DeviceBatch.from_host(base, off, lens, nb*BLOCK, (ar*BLOCK).astype(np.uint64), np.full(nb, BLOCK, dtype=np.uint64), dev)
        for v in (9, 10):
            ctx.set_variant(N.ZLIB_DECODE, v)
            sc = torch.empty(ctx.scratch_bytes(N.ZLIB_DECODE, nb, BLOCK) + 256, dtype=torch.uint8, device=dev)
            for _ in range(2): ctx.launch_dev(N.ZLIB_DECODE, db, sc)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): ctx.launch_dev(N.ZLIB_DECODE, db, sc)
            torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 3
            ok = int(db.status[:nb].abs().max()) == 0 and torch.equal(db.out_base[:nb*BLOCK].cpu(), torch.from_numpy(raw_np))
            print("members %6d x %7d B variant %d: %.2f ms  %.1f GiB/s ok=%s" % (nb, BLOCK, v, t*1e3, nb*BLOCK/t/2**30, ok), flush=True)
