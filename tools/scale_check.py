#!/usr/bin/env python
"""Multi-GPU day-one check (verdict r5 item 7): sharded results must be BIT-IDENTICAL to the one-GPU index.

  one process, the library's sharded handle (ls_create_sharded, RCCL inside the library):
      python tools/scale_check.py --gpus 1,2,4,8
  one process per GPU (torch.distributed, nccl = RCCL):
      python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 --master-port P \\
          tools/scale_check.py --dist
  rehearsal on fewer GPUs than shards (device ids repeat; collectives over gloo / copies):  --share

Shapes: c2 (N = 200 k, d = 384 fp32: nq = 1, 16, 32 on the exact scan / ls_mq path, k = 50 and 1000; nq = 200 on the
verified f32 MFMA path) and a c4-shaped fp16 corpus (d = 768, --c4-rows rows per shard, nq = 256, k = 100). Every
sharded answer is compared with `array_equal` - scores and rows - to the same search on ONE index over all the rows
(reference call: index.search(x, k), src/lean_explore/search/engine.py:250). Prints the handle's exchange record
(ls_shard_exchange_info: transport, RCCL version, peer matrix)."""
from __future__ import annotations

import argparse
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def gauss(seed, n, d):
    x = np.random.default_rng(seed).standard_normal((n, d), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


def cases(c2_rows, c4_rows, shards):
    c2 = gauss(1234, c2_rows, 384)
    q2 = gauss(5678, 200, 384)
    yield "c2 f32", c2, "f32", [(q2[:1], 50), (q2[:16], 50), (q2[:32], 50), (q2[:3], 1000), (q2, 50)]
    n4 = c4_rows * shards
    c4 = gauss(4321, n4, 768)
    yield f"c4-shape f16 ({c4_rows} rows per shard)", c4, "f16", [(gauss(8765, 256, 768), 100), (gauss(8766, 1, 768), 100)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", default="1,2,4,8")
    ap.add_argument("--dist", action="store_true", help="one process per GPU: run under torch.distributed.run")
    ap.add_argument("--share", action="store_true", help="rehearsal: shards share the visible GPU(s)")
    ap.add_argument("--c2-rows", type=int, default=200_000)
    ap.add_argument("--c4-rows", type=int, default=1_000_000)
    args = ap.parse_args()
    import torch

    from lean_explore_amd.index import FlatIPIndex

    ndev = torch.cuda.device_count()
    ok = True
    if args.dist:
        import torch.distributed as dist

        from lean_explore_amd.sharded import ShardedFlatIPIndex, shard_bounds

        world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
        local_rank = int(os.environ.get("LOCAL_RANK", rank))
        if world > ndev and not args.share:
            raise SystemExit(f"{world} ranks but {ndev} GPUs (use --share for a rehearsal)")
        dev_i = local_rank % max(1, ndev)
        torch.cuda.set_device(dev_i)
        if args.share and world > ndev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_i))
        for name, corpus, dtype, calls in cases(args.c2_rows, args.c4_rows, world):
            lo, hi = shard_bounds(corpus.shape[0], world, rank)
            local = FlatIPIndex.from_array(np.ascontiguousarray(corpus[lo:hi]), dtype=dtype, device=dev_i, base=lo)
            sh = ShardedFlatIPIndex(local, corpus.shape[0])
            ref = FlatIPIndex.from_array(corpus, dtype=dtype, device=dev_i) if rank == 0 else None
            for q, k in calls:
                D, I = sh.search(q, k)
                if rank == 0:
                    Dr, Ir = ref.search(q, k)
                    same = np.array_equal(D, Dr) and np.array_equal(I, Ir)
                    ok &= same
                    print(f"[dist x{world}] {name} nq={q.shape[0]} k={k}: {'bit-identical to one index' if same else 'DIFFERS'}", flush=True)
            local.close()
            if ref is not None:
                ref.close()
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            print("scale_check dist:", "OK" if ok else "FAILED", flush=True)
        sys.exit(0 if ok else 1)

    for G in [int(x) for x in args.gpus.split(",")]:
        if G > ndev and not args.share:
            print(f"[inlib x{G}] skipped: {ndev} GPUs visible", flush=True)
            continue
        devs = [g % max(1, ndev) for g in range(G)]
        for name, corpus, dtype, calls in cases(args.c2_rows, args.c4_rows, G):
            ref = FlatIPIndex.from_array(corpus, dtype=dtype, device=0)
            ix = FlatIPIndex.from_array(corpus, dtype=dtype, devices=devs) if G > 1 else ref
            for q, k in calls:
                D, I = ix.search(q, k)
                Dr, Ir = ref.search(q, k)
                same = np.array_equal(D, Dr) and np.array_equal(I, Ir)
                ok &= same
                print(f"[inlib x{G}] {name} nq={q.shape[0]} k={k}: {'bit-identical to one index' if same else 'DIFFERS'}", flush=True)
            if G > 1:
                print(f"[inlib x{G}] exchange: " + json.dumps(ix.exchange_info()), flush=True)
                ix.close()
            ref.close()
    print("scale_check inlib:", "OK" if ok else "FAILED", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
