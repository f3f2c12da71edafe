#!/usr/bin/env python3
"""bench.py -- queries/sec of the hot path: DenseNet-121 embed (224x224) + exact top-10 over a
resident 1M x 1024 gallery (BASELINE.json metric), on N GPUs of one node.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus 8 --steps 20 --warmup 5          (starts its own 8 ranks: self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" on every rank: embed `--queries` synthetic images (micro-batches of
`--embed-batch`, already resident in HBM), all-gather the embeddings (N > 1), exact top-10 of
ALL N*queries embeddings against the local gallery shard (1M/N rows; bf16 MFMA candidates +
fp64 re-rank in libmirx), one all-gather of the per-shard candidates, merge.  Per-GPU work is
constant in N (weak scaling): value = N * queries * steps / max-over-ranks time.

`--embed-streams S` splits a micro-batch over S HIP streams (the ~250 dependent kernels of a forward leave tails that
another stream's forward fills).  For N > 1 the line carries per-stage times (embed / gather_q / search / gather_cand /
merge, rank 0 and the maximum over ranks) so that a scaling curve can be read stage by stage.

Extra objects on the JSON line (tier contract):
  roofline      the distance GEMM (k_gemm filter pass): algorithmic 2*Q*N_shard*D FLOP per
                launch / its HIP-event duration measured inside the timed steps, vs the dense
                bf16 MFMA peak 2516.6 TFLOP/s (256 CU x 2.4 GHz x 4096 FLOP/clk/CU).
  roofline_embed  the embed stage's dominant kernel family (the fused 1x1 convolutions, two fp16 terms): HBM-bound --
                algorithmic bytes (each launch reads its input prefix once and writes its output once) / HIP-event time,
                vs 8 TB/s; `traffic` = FETCH_SIZE x 2 + WRITE_SIZE from the committed PMC passes of the same forward.
  extras        (N = 1) images/s of the other backbones at their native resolutions (ConvNeXtV2-base 384, DINOv2 ViT-B/14
                518, MedSigLIP 448) and search-only queries/s over 1M x 256 / 512 galleries, measured in this run.
  cpu_baseline  the reference's CPU path re-created with the same torch calls
                (`-torch.cdist` + topk, test.py:1080,44) and the oracle DenseNet restatement,
                timed on this box's host cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL / tensor sharing fail with the legacy mode); the driver's
# environment exports it already -- this only covers a bare shell
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2516.6
MFMA_F32_PEAK_TFLOPS = 157.3          # v_mfma_f32_32x32x2_f32: 64 FLOP/clk/SIMD (guide, chip-level table)
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gallery", type=int, default=1_000_000, help="total gallery rows (all GPUs)")
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--queries", type=int, default=4096, help="queries (images) per GPU per step")
    ap.add_argument("--embed-batch", type=int, default=4096)
    ap.add_argument("--embed-streams", type=int, default=2, help="HIP streams one micro-batch is split over")
    ap.add_argument("--search-chunks", type=int, default=1, help="N > 1: pieces of the local search whose candidate all-gathers overlap the next piece")
    ap.add_argument("--no-extras", action="store_true", help="skip the other-backbone / other-dimension measurements")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--search-only", action="store_true", help="skip the embed stage (dev aid; not the metric)")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="ranks only rendezvous (gloo) and print a line: exercises the self-launch path without a GPU")
    return ap.parse_args()


def synthetic_images(batch, size, seed, device):
    """torch.rand in [0,1) then ImageNet normalisation (test.py:1309-1310), on device."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.rand((batch, 3, size, size), generator=g, device=device)
    mean = torch.tensor(IMAGENET_MEAN, device=device).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=device).view(1, 3, 1, 1)
    return (x - mean) / std


def gallery_chunk(chunk_idx, rows, dim, device):
    g = torch.Generator(device=device).manual_seed(1234 + chunk_idx)
    return torch.nn.functional.normalize(torch.randn((rows, dim), generator=g, device=device), dim=1)


def build_shard(index, lo, hi, dim, device, chunk=1 << 16):
    """Rows [lo, hi) of the seeded synthetic gallery; chunking is global so shards tile it."""
    index.reserve(hi - lo)
    c0, c1 = lo // chunk, (hi - 1) // chunk if hi > lo else -1
    for c in range(c0, c1 + 1):
        rows = gallery_chunk(c, chunk, dim, device)
        a, b = max(lo, c * chunk), min(hi, (c + 1) * chunk)
        index.add(rows[a - c * chunk: b - c * chunk], torch.arange(a, b, device=device))


def cpu_baseline(index, model, args, dev):
    """Reference CPU path on a bounded sample: embed 64 images (B=64, test.py:1513) + cdist/topk of 32 queries."""
    from oracle import densenet as OD
    nthreads = torch.get_num_threads()
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    imgs = synthetic_images(64, args.image_size, 777, dev).cpu()
    with torch.no_grad():
        OD.embed(imgs[:2], sd)
        t0 = time.perf_counter()
        emb = OD.embed(imgs, sd)
        t_embed = (time.perf_counter() - t0) / imgs.shape[0]
    g_host = torch.empty((len(index), args.dim), dtype=torch.float32)
    step = 1 << 17
    for s in range(0, len(index), step):
        rows, _ = index.rows(s, min(step, len(index) - s))
        g_host[s:s + rows.shape[0]] = rows.cpu()
    q = emb[:32].contiguous()                      # the embeddings just computed are the queries
    with torch.no_grad():
        t0 = time.perf_counter()
        d = -torch.cdist(q, g_host)
        d.topk(args.k, 1, True, True)
        t_search = (time.perf_counter() - t0) / q.shape[0]
    return {"value": 1.0 / (t_embed + t_search), "unit": "queries/s", "cores": nthreads, "kind": "port",
            "sample": f"64 images embedded (oracle DenseNet-121 fp32, {1.0 / t_embed:.1f} img/s) + 32 queries "
                      f"-torch.cdist+topk({args.k}) over {len(index)}x{args.dim} fp32 ({1.0 / t_search:.1f} q/s); "
                      f"torch CPU threads={nthreads}, os.cpu_count={os.cpu_count()}"}


def profile_traffic(name, key):
    """Fabric-side bytes from the committed PMC summary of the newest round that has one (tools/pmc_traffic.py: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH x 2 on gfx950).  -> (value, file name) or (None, None)."""
    for rnd in ("r04", "r03"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json")
        if os.path.exists(path):
            with open(path) as fh:
                d = json.load(fh)
            v = d.get(name, {}).get(key)
            if v is not None:
                return v, os.path.basename(path)
    return None, None


# ---- algorithmic work of the other backbones (per image): every Linear / conv reads its input once and writes its output once
# (fp32), elementwise steps (LayerNorm, GELU, GRN, residual, softmax) count as fused; FLOP = matrix FLOP.  The rule of SURVEY 8d.
def convnextv2_work(size=384, dims=(128, 256, 512, 1024), depths=(3, 3, 27, 3)):
    t = (size // 4) ** 2
    flop, elems = 2.0 * t * 48 * dims[0], t * (48 + dims[0])                       # stem 4x4 / 4
    for i, (c, d) in enumerate(zip(dims, depths)):
        if i:
            flop += 2.0 * (t // 4) * 4 * dims[i - 1] * c                           # downsample 2x2 / 2
            elems += t * dims[i - 1] + (t // 4) * c
            t //= 4
        flop += d * (2.0 * 49 * c * t + 16.0 * t * c * c)                          # dwconv 7x7 + fc1 + fc2 (hidden 4 c)
        elems += d * 12 * t * c                                                    # dw in/out, fc1 in + 4c out, fc2 4c in + out
    return flop, 4.0 * elems


def vit_work(tokens, c, hidden, depth, patch_k):
    flop = 2.0 * tokens * patch_k * c + depth * (2.0 * tokens * c * (4 * c + 2 * hidden) + 4.0 * tokens * tokens * c)
    elems = tokens * (patch_k + c) + depth * (10 * tokens * c + 2 * tokens * (c + hidden))      # qkv, attention, proj, fc1, fc2
    return flop, 4.0 * elems


FP32_EQ_PEAK_TFLOPS = 2516.6 / 3.0        # two fp16 terms per operand: three dense fp16 MFMAs per fp32-grade product


def timed_images_per_s(model, x, iters, warmup=2):
    with torch.no_grad():
        for _ in range(warmup):
            model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            model(x)
        torch.cuda.synchronize()
    return x.shape[0] * iters / (time.perf_counter() - t0)


def densenet_batch_curve(model, args, dev, batches=(1, 8, 32, 64, 256, 1024)):
    """The reference's loops run at B = 64 (test.py:1513), 32 (ingest_embeddings.py:467), 1 (milvus_retrieval.py:53-66) and
    ~53 per metric call (the XAI loop): images/s and ms per forward at those batch sizes, ONE stream, device-resident fp32
    images.  The step itself embeds 4096 images on two streams (`value`)."""
    out = []
    x = synthetic_images(max(batches), args.image_size, 4242, dev)
    with torch.no_grad():
        for b in batches:
            xb = x[:b].contiguous()
            for _ in range(3):
                model(xb)
            torch.cuda.synchronize(dev)
            iters = max(3, min(200, int(2000 / max(b, 8))))
            t0 = time.perf_counter()
            for _ in range(iters):
                model(xb)
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / iters
            out.append({"batch": b, "ms_per_forward": dt * 1e3, "images_per_s": b / dt})
    return out


def overlap_ab(model, index, pool, args, dev, steps=6):
    """search of step i beside the embed of step i + 1 (mirx_index_search_begin / _end on its own stream, two embedding
    buffers) against the plain sequence, same box, same run: queries/s of both.  N = 1 only."""
    q = pool[0].shape[0]
    embs = [torch.empty((q, args.dim), dtype=torch.float32, device=dev) for _ in range(2)]
    side = [torch.cuda.Stream(device=dev) for _ in range(2)]
    sstream = torch.cuda.Stream(device=dev)
    cur = torch.cuda.current_stream(dev)
    done = [None, None]                  # event on the search stream: every pass that reads embs[i] has been enqueued and run

    def embed(dst):
        part = q // 2
        for j, st in enumerate(side):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                dst[j * part:(j + 1) * part] = model(pool[0][j * part:(j + 1) * part])
        for st in side:
            cur.wait_stream(st)

    def run(overlap):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        pending = None
        with torch.no_grad():
            for i in range(steps):
                buf = embs[i % 2]
                if overlap and done[i % 2] is not None:
                    cur.wait_event(done[i % 2])          # search_end(i - 2)'s follow-up passes read this buffer
                embed(buf)
                if not overlap:
                    index.search(buf, args.k, return_f64=True)
                    continue
                sstream.wait_stream(cur)
                if pending is not None:
                    with torch.cuda.stream(sstream):
                        index.search_end(pending, return_f64=True)
                        done[(i - 1) % 2] = torch.cuda.Event()
                        done[(i - 1) % 2].record(sstream)
                with torch.cuda.stream(sstream):
                    pending = index.search_begin(buf, args.k)
            if pending is not None:
                with torch.cuda.stream(sstream):
                    index.search_end(pending, return_f64=True)
        torch.cuda.synchronize(dev)
        return steps * q / (time.perf_counter() - t0)

    run(False)
    off = run(False)
    run(True)
    on = run(True)
    return {"off_queries_per_s": off, "on_queries_per_s": on, "steps": steps, "queries_per_step": q,
            "note": "on: the search of step i runs on its own stream beside the embed of step i + 1 (search_begin/_end)"}


def host_resident_inputs(model, index, args, dev, steps=4):
    """The reference's loop hands HOST batches to the device (test.py:1070-1075).  Here: 8-bit images in pinned host memory
    -> device on a copy stream (double-buffered, overlapped with the previous batch's embed + search) -> uint8 stem.  The
    PCIe-inclusive queries/s of the whole step; fp32 host images would move four times the bytes."""
    q = args.embed_batch
    host = [torch.randint(0, 256, (q, 3, args.image_size, args.image_size), dtype=torch.uint8).pin_memory() for _ in range(2)]
    devb = [torch.empty((q, 3, args.image_size, args.image_size), dtype=torch.uint8, device=dev) for _ in range(2)]
    copy = torch.cuda.Stream(device=dev)
    cur = torch.cuda.current_stream(dev)
    side = [torch.cuda.Stream(device=dev) for _ in range(2)]
    emb = torch.empty((q, args.dim), dtype=torch.float32, device=dev)
    ready = [torch.cuda.Event() for _ in range(2)]
    consumed = [torch.cuda.Event() for _ in range(2)]

    def upload(i):
        with torch.cuda.stream(copy):
            copy.wait_event(consumed[i % 2])
            devb[i % 2].copy_(host[i % 2], non_blocking=True)
            ready[i % 2].record(copy)

    def one(i):
        cur.wait_event(ready[i % 2])
        part = q // 2
        for j, st in enumerate(side):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                emb[j * part:(j + 1) * part] = model(devb[i % 2][j * part:(j + 1) * part])
        for st in side:
            cur.wait_stream(st)
        consumed[i % 2].record(cur)
        index.search(emb, args.k, return_f64=True)

    with torch.no_grad():
        for ev in consumed:
            ev.record(cur)
        upload(0)
        one(0)                                                 # warm-up step (uint8 path, pinned buffers)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        upload(0)
        for i in range(steps):
            if i + 1 < steps:
                upload(i + 1)
            one(i)
        torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    return {"queries_per_s": q / dt, "ms_per_step": dt * 1e3, "h2d_mb_per_step": q * 3 * args.image_size ** 2 / 1e6,
            "input": "uint8 [B, 3, H, W] in pinned host memory, normalised inside the stem kernel", "steps": steps}


def retriever_single_query(model, args, dev, rows=200_000, calls=40):
    """The reference's per-query loop (milvus_retrieval.py:53-86, evaluate_test_dataset_milvus.py:446-460): ONE PIL image ->
    transform -> model(img[None]) -> top-k, through mirx.retriever.MilvusRetriever.search.  queries/s and where the time goes
    (the B = 1 forward is 125 dependent launches: DESIGN 11)."""
    import numpy as np
    from PIL import Image
    from mirx.retriever import MilvusManager, MilvusRetriever, default_transform
    mgr = MilvusManager(device=dev)
    mgr.connect()
    mgr.create_collection("densenet121", drop_old=True)
    col = mgr.collections["densenet121"]
    g = torch.Generator(device=dev).manual_seed(7)
    for s0 in range(0, rows, 50_000):
        n = min(50_000, rows - s0)
        emb = torch.nn.functional.normalize(torch.randn((n, args.dim), generator=g, device=dev), dim=1)
        col.insert([[f"img_{s0 + i}.png" for i in range(n)], ["normal"] * n, emb])
    r = MilvusRetriever(mgr, "densenet121", model, default_transform(args.image_size))
    r.load_collection()
    img = Image.fromarray(np.random.default_rng(0).integers(0, 256, (300, 280, 3), dtype=np.uint8))
    for _ in range(3):
        r.search(img, top_k=args.k)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(calls):
        r.search(img, top_k=args.k)
    torch.cuda.synchronize(dev)
    per = (time.perf_counter() - t0) / calls
    x = r.transform(img).unsqueeze(0).to(dev)
    t1 = time.perf_counter()
    for _ in range(calls):
        r.transform(img)
    t_tf = (time.perf_counter() - t1) / calls
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    with torch.no_grad():
        for _ in range(calls):
            model(x)
    torch.cuda.synchronize(dev)
    t_fw = (time.perf_counter() - t2) / calls
    return {"queries_per_s": 1.0 / per, "ms_per_query": per * 1e3, "ms_transform_cpu": t_tf * 1e3, "ms_forward_b1": t_fw * 1e3,
            "gallery_rows": rows, "k": args.k, "note": "MilvusRetriever.search(PIL image): resize + crop + normalise on the host, "
            "B = 1 forward, exact top-k, result dicts"}


def extras(args, dev):
    """Configs 3-5 of BASELINE.json on this GPU (embed stage, native resolution, fp32, synthetic images, random-init
    weights) and the search stage at their embedding widths over a 1M-row gallery."""
    from mirx import _lib
    from mirx.index import FlatIndex
    from mirx.model import ConvNeXtV2, DinoV2, MedSigLIP
    out = {}
    work = {"convnextv2_base_384": convnextv2_work(384), "dinov2_vitb14_518": vit_work(37 * 37 + 1, 768, 3072, 12, 588),
            "medsiglip_448": vit_work(32 * 32, 1152, 4304, 27, 588)}
    for name, ctor, size, batch, iters in (("convnextv2_base_384", lambda: ConvNeXtV2(embedding_dim=256), 384, 64, 4),
                                           ("dinov2_vitb14_518", lambda: DinoV2(embedding_dim=256), 518, 32, 4),
                                           ("medsiglip_448", lambda: MedSigLIP(embed_dim=512), 448, 16, 4)):
        torch.manual_seed(0)
        m = ctor().eval().to(dev)
        x = synthetic_images(batch, size, 99, dev)
        ips = timed_images_per_s(m, x, iters)
        one = 1e3 / timed_images_per_s(m, x[:1].contiguous(), 10, warmup=3)        # the reference's per-query loop: one image per call
        flop, nbytes = work[name]
        tr, tr_file = profile_traffic(name, "bytes_per_image")
        tr_batch, _ = profile_traffic(name, "batch")
        f_frac, b_frac = flop * ips / 1e12 / FP32_EQ_PEAK_TFLOPS, nbytes * ips / 1e9 / 8000.0
        out[name] = {"images_per_s": ips, "batch": batch, "single_image_ms": one, "image_size": size, "dtype": "f32 (two fp16 / three bf16 MFMA terms)",
                     "data": "synthetic images, random-init weights",
                     "algorithmic_gflop_per_image": flop / 1e9, "algorithmic_mb_per_image": nbytes / 1e6,
                     "roofline": {"bound": "mfma" if f_frac >= b_frac else "hbm", "frac_of_fp32_equivalent_mfma_peak": f_frac,
                                  "frac_of_hbm_peak": b_frac, "frac": max(f_frac, b_frac),
                                  "peaks": f"{FP32_EQ_PEAK_TFLOPS:.1f} TFLOP/s (dense fp16 MFMA / 3 terms), 8000 GB/s"},
                     "traffic_bytes_per_image": tr, "traffic_profile": tr_file, "traffic_profile_batch": tr_batch,
                     "dominant_kernel": profile_traffic(name, "dominant_kernel")[0]}
        del m, x
        torch.cuda.empty_cache()
    for dim in (256, 512):
        ix = FlatIndex(dim, "COSINE", dev.index)
        build_shard(ix, 0, args.gallery, dim, dev)
        ix.set_option(_lib.OPT_PROFILE, 1)
        q = torch.nn.functional.normalize(
            torch.randn((4096, dim), generator=torch.Generator(device=dev).manual_seed(4321), device=dev), dim=1)
        ix.search(q, args.k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ix.search(q, args.k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        gm = ix.last_timings()["gemm"]
        dimp = (dim + 127) // 128 * 128
        tf = 2.0 * 4096 * args.gallery * dimp / (gm * 1e-3) / 1e12 if gm > 0 else 0.0
        out[f"search_only_{args.gallery}x{dim}"] = {"queries_per_s": 4096 / dt, "ms_per_4096_queries": dt * 1e3, "gemm_ms": gm,
                                                    "gemm_tflops": tf, "gemm_frac_of_bf16_peak": tf / MFMA_BF16_PEAK_TFLOPS}
        del ix
        torch.cuda.empty_cache()
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes through torch.distributed.run
    (127.0.0.1 rendezvous on a free port) BEFORE this process makes any GPU call, forward rank 0's JSON line and
    return the children's exit code.  A child process, never an exec: this pool forbids replacing a process."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env["MIRX_BENCH_CHILD"] = "1"
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = []
    for ln in proc.stdout:
        if ln.lstrip().startswith("{") and '"metric"' in ln:
            lines.append(ln.strip())
        else:
            sys.stderr.write(ln)           # launcher chatter stays off the contract's stdout
    rc = proc.wait()
    if rc == 0 and len(lines) != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        rc = 3
    for ln in lines[-1:]:
        print(ln, flush=True)
    return rc


def rank_provenance(world, rank, local_rank, rehearse, device_name):
    """What proves, on the line itself, which backend carried the collectives and which device every rank drove:
    gathered from all ranks (all_gather_object runs on the default group's backend)."""
    import socket
    import torch.distributed as dist
    mine = {"rank": rank, "local_rank": local_rank, "device": device_name, "host": socket.gethostname(), "pid": os.getpid(),
            "visible_gpus": torch.cuda.device_count()}
    if world == 1:
        return {"collective_backend": None, "rehearsal": False, "ranks": [mine]}
    allr = [None] * world
    dist.all_gather_object(allr, mine)
    return {"collective_backend": dist.get_backend(), "rehearsal": bool(rehearse), "ranks": allr}


def refuse_rehearsal_on_a_multi_gpu_node(args):
    """MIRX_BENCH_REHEARSE=1 (all ranks on cuda:0, collectives over gloo) is a one-GPU development aid.  On a node with more
    than one visible GPU a --gpus N > 1 run with it set would print a line that LOOKS like a scaling point: refuse."""
    if os.environ.get("MIRX_BENCH_REHEARSE") == "1" and args.gpus > 1 and torch.cuda.device_count() > 1:
        print("bench.py: MIRX_BENCH_REHEARSE=1 with --gpus > 1 on a node with more than one visible GPU: refusing to "
              "produce a rehearsal line where a real RCCL run is possible", file=sys.stderr)
        sys.exit(4)


def launch_selftest(world, rank):
    """--selftest-launch: what a rank does instead of the GPU bench when only the launch path is under test (CPU
    container): gloo rendezvous, one all-reduce, rank 0 prints a line of the contract's shape."""
    import torch.distributed as dist
    if os.environ.get("MIRX_BENCH_SELFTEST_FAIL") == str(rank):
        sys.exit(7)                                    # a rank that dies: the launcher must report it
    if world > 1:
        dist.init_process_group("gloo")
    t = torch.ones(1, dtype=torch.float64) * (rank + 1)
    if world > 1:
        dist.all_reduce(t)
    prov = rank_provenance(world, rank, int(os.environ.get("LOCAL_RANK", "0")), False, "cpu (launch selftest)")
    if rank == 0:
        print(json.dumps({"metric": "launch selftest", "value": float(t.item()), "unit": "sum of ranks + 1", "n_gpus": world,
                          "launch_selftest": True, **prov}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.selftest_launch and os.environ.get("MIRX_BENCH_SELFTEST_VISIBLE_GPUS"):   # test hook: pretend the node shows that many GPUs
        torch.cuda.device_count = lambda: int(os.environ["MIRX_BENCH_SELFTEST_VISIBLE_GPUS"])
    refuse_rehearsal_on_a_multi_gpu_node(args)       # before any rank is started (device_count() does not initialise the GPU)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1 and os.environ.get("MIRX_BENCH_CHILD") != "1":
            sys.exit(self_launch(args))
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if args.selftest_launch:
        launch_selftest(world, rank)
        return
    import torch.distributed as dist
    from mirx import _lib
    from mirx.dist import ShardedSearcher, shard_bounds
    from mirx.index import FlatIndex
    from mirx.model import DenseNet121

    # MIRX_BENCH_REHEARSE=1: development rehearsal of the N>1 control flow on a ONE-GPU box (all ranks
    # share cuda:0, collectives over gloo).  Never used for reported numbers.
    rehearse = os.environ.get("MIRX_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    # ---- resident state ------------------------------------------------------------------
    torch.manual_seed(0)
    model = DenseNet121().eval().to(dev)
    index = FlatIndex(args.dim, "COSINE", local_rank)
    lo, hi = shard_bounds(args.gallery, world, rank)
    build_shard(index, lo, hi, args.dim, dev)
    index.set_option(_lib.OPT_PROFILE, 1)
    args.embed_batch = min(args.embed_batch, args.queries)
    nmb = max(1, args.queries // args.embed_batch)
    q_local = nmb * args.embed_batch
    pool = [synthetic_images(args.embed_batch, args.image_size, 1234 + 97 * rank + i, dev) for i in range(min(2, nmb))]
    fixed_q = torch.nn.functional.normalize(
        torch.randn((q_local, args.dim), generator=torch.Generator(device=dev).manual_seed(4321 + rank), device=dev), dim=1)
    searcher = ShardedSearcher(lambda qa, kk: index.search(qa, kk, return_f64=True), "COSINE")
    emb = torch.empty((q_local, args.dim), dtype=torch.float32, device=dev)
    gemm_ms, stage_ms, stage_ev = [], {}, {}
    nstr = max(1, args.embed_streams)
    side = [torch.cuda.Stream(device=dev) for _ in range(nstr)] if nstr > 1 and args.embed_batch % nstr == 0 else None

    def embed_all():
        """Every micro-batch through the DenseNet; with side streams its halves run concurrently (each forward is a chain
        of ~250 dependent kernels whose tails the other stream's kernels fill)."""
        cur = torch.cuda.current_stream(dev)
        for i in range(nmb):
            x = pool[i % len(pool)]
            dst = emb[i * args.embed_batch:(i + 1) * args.embed_batch]
            if side is None:
                dst.copy_(model(x))
                continue
            part = args.embed_batch // len(side)
            for j, st in enumerate(side):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    dst[j * part:(j + 1) * part] = model(x[j * part:(j + 1) * part])
            for st in side:
                cur.wait_stream(st)

    def step(record):
        ev = stage_ev if record else None
        with torch.no_grad():
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            if args.search_only:
                emb.copy_(fixed_q)
            else:
                embed_all()
            e1.record()
            if record:
                stage_ev.setdefault("embed", []).extend([(True, e0), (False, e1)])
            out = searcher.search(emb, args.k, chunks=args.search_chunks if world > 1 else 1, events=ev)
        if record:
            t = index.last_timings()          # waits for this search's events only
            gemm_ms.append(t["gemm"])
            for kname, v in t.items():
                stage_ms[kname] = stage_ms.get(kname, 0.0) + v
        return out

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-stage times of the timed steps (events recorded on the compute stream; an asynchronous collective shows
    # its launch-side time only) ------------------------------------------------------------------------
    names = ["embed", "gather_q", "search", "gather_cand", "merge"]
    mine = []
    for nm in names:
        pairs = stage_ev.get(nm, [])
        tot, begin = 0.0, None
        for is_begin, ev in pairs:
            if is_begin:
                begin = ev
            elif begin is not None:
                tot += begin.elapsed_time(ev)
                begin = None
        mine.append(tot / max(1, args.steps))
    stage_rank0, stage_max = mine, mine
    if world > 1:
        tt = torch.tensor(mine, dtype=torch.float64, device=dev)
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.broadcast(tt, src=0)
        stage_rank0, stage_max = tt.tolist(), mx.tolist()

    total_q = world * q_local * args.steps
    stats = index.last_stats()
    prov = rank_provenance(world, rank, local_rank, rehearse, torch.cuda.get_device_name(dev))
    embed_roof = None
    if rank == 0 and not args.search_only:
        # calibration pass OUTSIDE the timed region: HIP events around every launch of the embed stage's dominant
        # hand-written kernel family (the fused 1x1 convolutions of the 58 dense layers and 3 transitions)
        # Two passes.  (1) ONE stream, the images one stream of the step embeds: a kernel then has the chip to itself and its
        # duration is its own -- this is `achieved` (and what the committed one-stream rocprof summary must agree with).
        # (2) the step's own configuration, the micro-batch split over the embed streams: launches of the two streams overlap,
        # so a launch's HIP-event duration includes the time it shares the chip -- reported beside it as `in_step`.
        per_stream = args.embed_batch // (len(side) if side else 1)
        xcal = pool[0][:per_stream]

        def families():
            out = {}
            for ev_a, ev_b, f, nb, kind in model.conv1x1_timer:
                e = out.setdefault(kind, [0.0, 0.0, 0.0, 0])
                e[0] += ev_a.elapsed_time(ev_b)
                e[1] += f
                e[2] += nb
                e[3] += 1
            return out

        in_step = None
        if side is not None and nmb == 1:
            model.conv1x1_timer = []
            with torch.no_grad():
                embed_all()
            torch.cuda.synchronize(dev)
            f2 = families()
            if f2:
                _, (ms2, _, nb2, nl2) = max(f2.items(), key=lambda kv: kv[1][0])
                in_step = {"streams": len(side), "launches": nl2, "sum_of_launch_ms": ms2, "algorithmic_bytes": nb2,
                           "GBps_over_summed_launch_time": nb2 / (ms2 * 1e-3) / 1e9,
                           "note": "HIP-event durations of launches that share the chip with the other stream's kernels"}
        model.conv1x1_timer = []
        with torch.no_grad():
            model(xcal)
        torch.cuda.synchronize(dev)
        fams = families()
        model.conv1x1_timer = None
        if fams:
            # the dominant hand-written family of the embed stage by time (the fused 1x1 convolutions of the 58 dense layers and
            # 3 transitions).  Algorithmic bytes: a launch reads cin channels and writes cout channels of every pixel once, fp32.
            kind, (ms, fl, nbytes, nl) = max(fams.items(), key=lambda kv: kv[1][0])
            b = xcal.shape[0]
            gbs = nbytes / (ms * 1e-3) / 1e9
            tr, tr_file = profile_traffic("densenet121_conv1x1", "bytes_per_image")
            embed_roof = {"bound": "hbm",
                          "kernel": ("mirx::k_conv1x1_h2 (fused BN+ReLU+1x1 conv+BN+ReLU, two fp16 MFMA terms, fp32-grade), "
                                     if kind == "conv1x1" else "mirx::k_dense_fused (one-launch dense layer, bottleneck in LDS), ")
                                    + f"{nl} launches of one {b}-image forward on one stream",
                          "dtype": "f32 (2 x fp16 terms)", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                          "algorithmic_bytes_per_forward": nbytes, "ms_per_forward": ms,
                          "fp32_equivalent_tflops": fl / (ms * 1e-3) / 1e12,
                          "traffic": None if tr is None else tr * b, "traffic_profile": tr_file, "in_step": in_step}
    if rank == 0:
        dimp = (args.dim + 63) // 64 * 64
        flop = 2.0 * (world * q_local // max(1, args.search_chunks if world > 1 else 1)) * (hi - lo) * dimp
        avg_gemm_ms = sum(gemm_ms) / max(1, len(gemm_ms))
        achieved = flop / (avg_gemm_ms * 1e-3) / 1e12 if avg_gemm_ms > 0 else 0.0
        line = {
            "metric": "queries/sec (embed+top-10) over 1M x 1024 gallery" if not args.search_only else "queries/sec (search only)",
            "value": total_q / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "collective_backend": prov["collective_backend"],       # "nccl" (= RCCL) for N > 1; null for one GPU
            "rehearsal": prov["rehearsal"],
            "ranks": prov["ranks"],
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 embed (two fp16 MFMA terms); bf16 MFMA candidates + f64 re-rank",
            "data": "synthetic (rand 224x224 ImageNet-normalised images, seed-0 random-init DenseNet-121, "
                    "unit-norm randn gallery seed 1234)",
            "config": {"workload": f"DenseNet-121 embed {args.image_size}x{args.image_size} (device-resident inputs) + exact "
                                   f"top-{args.k} over {args.gallery}x{args.dim} fp32 gallery (cosine), {q_local} queries/GPU/step",
                       "gallery_rows": args.gallery, "dim": args.dim, "queries_per_gpu_per_step": q_local,
                       "embed_batch": args.embed_batch, "embed_streams": len(side) if side else 1, "k": args.k,
                       "sharding": (f"gallery rows / {world} GPUs + 2 all-gathers, search in {args.search_chunks} piece(s)"
                                    if world > 1 else "single GPU")
                                   + (" [REHEARSAL: ranks share one GPU, gloo]" if rehearse else ""),
                       "search_stats_last_step": stats,
                       "stage_ms_per_step": {k: v / args.steps for k, v in stage_ms.items()},
                       "pipeline_ms_per_step_rank0": dict(zip(names, stage_rank0)),
                       "pipeline_ms_per_step_max_over_ranks": dict(zip(names, stage_max))},
            "roofline": {"bound": "mfma", "kernel": "mirx::k_gemm16<0,false> (bf16 16x16x32 MFMA distance GEMM + threshold filter)",
                         "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_BF16_PEAK_TFLOPS, "traffic": None,
                         "flop_per_launch": flop, "avg_launch_ms": avg_gemm_ms},
        }
        if embed_roof is not None:
            line["roofline_embed"] = embed_roof
        if world == 1 and q_local == 4096 and args.gallery == 1_000_000 and args.dim == 1024:
            tr, tr_file = profile_traffic("gemm_1Mx1024_q4096", "bytes_per_launch")
            line["roofline"]["traffic"] = tr
            line["roofline"]["traffic_profile"] = tr_file
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(index, model, args, dev)
        if world == 1 and not args.no_extras and not args.search_only:
            ab = overlap_ab(model, index, pool, args, dev) if args.embed_batch % 2 == 0 else None
            hri = host_resident_inputs(model, index, args, dev) if args.embed_batch % 2 == 0 and args.image_size == 224 else None
            rsq = retriever_single_query(model, args, dev) if args.dim == 1024 else None
            curve = densenet_batch_curve(model, args, dev) if args.image_size == 224 else None
            del model, index, searcher, pool
            torch.cuda.empty_cache()
            line["extras"] = extras(args, dev)
            if ab is not None:
                line["extras"]["overlap_search_with_embed"] = ab
            if hri is not None:
                line["extras"]["host_resident_inputs"] = hri
            if rsq is not None:
                line["extras"]["retriever_single_query"] = rsq
            if curve is not None:
                line["extras"]["densenet_batch_curve"] = curve
            line["roofline"]["other_dims"] = {k: {kk: v[kk] for kk in ("gemm_ms", "gemm_tflops", "gemm_frac_of_bf16_peak")}
                                              for k, v in line["extras"].items() if k.startswith("search_only_")}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
