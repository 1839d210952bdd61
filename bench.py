#!/usr/bin/env python
"""bench.py -- MaxSim (query, doc) pairs scored per second on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  Rank 0 prints ONE JSON line.
N > 1: one process per GPU over RCCL.  Either the caller launches the ranks (torch.distributed.run: WORLD_SIZE
/ RANK / LOCAL_RANK in the environment) or -- when `--gpus N` is given without a launcher -- this file re-executes
itself under `python -m torch.distributed.run --nproc-per-node N` (launch_ranks below).  It refuses to run N ranks
on fewer than N visible GPUs (exit code 2) unless BENCH_SHARE_GPU=1 is set, which is a plumbing check, never a
measurement.  The N > 1 line carries `rccl_ranks` (an all-reduce of 1 over the process group) and every rank's
kernel time, so a run that silently degraded to fewer ranks cannot pass for a scaling point.

Workload (BASELINE.json: "MaxSim (query,doc) pairs scored/sec; achieved HBM GB/s vs peak",
synthetic 32-token-query x 1024-patch-doc x d=128): every rank holds a resident shard of a
pre-embedded corpus (packed bf16 blob in HBM, generated on the device); one step scores a query
batch against the whole shard with the fused gfx950 MaxSim kernel, selects the per-shard top-k
and (N>1) merges the shards' top-k with one RCCL all-gather.  Weak scaling: the shard per GPU is
fixed, the corpus grows with N.

stdout is ONE compact strict-JSON line (bench_legs/line.py: the contract keys, `roofline`, `cpu_baseline`, `topk_parity` and a flat
dict of scalar summaries, <= 8 KiB, no NaN tokens); the full report of every leg is written to bench_detail.json next to this file
(and to gpurun_out/ when that directory exists).  Nothing else is printed to stdout; stderr carries library banners only.

The JSON line carries
  roofline     -- of the dominant kernel (the fused MaxSim kernel), from HIP events recorded on the
                  launch stream inside the timed region; algorithmic bytes = docs streamed once
                  per query block (262144/Bq + 4 B per pair)
  cpu_baseline -- the reference's CPU scorer (oracle/torch_port.py restates
                  processing_utils.py:163-186 with the same torch calls) timed on this box's
                  host cores on a bounded sample of the same workload, rank 0, N=1 only
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench_legs.common import (HBM_PEAK_GBS, MFMA_PEAK_TFLOPS, make_queries, make_query_list, make_ragged_shard, make_shard,  # noqa: F401,E402
                               parse_regime, pmc_traffic, regime_numbers)

# The legs live in bench_legs/ and are imported when they run (tools/*.py reach them as bench.<name> through __getattr__ below).
_LEGS = {"resident": ("run_regime", "power_sample", "forced_collective_numbers", "stream_ceiling", "mfma_ceiling", "topk_parity",
                      "ragged_docs_numbers", "short_docs_numbers", "wide_320_numbers"),
         "baselines": ("cpu_baseline", "reference_scorer", "torch_gpu_reference"),
         "dropin": ("dropin_numbers", "embed_and_score_numbers"),
         "head": ("embed_head_numbers",),
         "loss_step": ("loss_step_numbers",),
         "vlm": ("vlm_in_the_loop_numbers",)}


def __getattr__(name):
    import importlib

    for mod, names in _LEGS.items():
        if name in names:
            return getattr(importlib.import_module(f"bench_legs.{mod}"), name)
    raise AttributeError(name)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs", type=int, default=125000,
                    help="documents per GPU shard (1024 patches x 128 bf16 = 256 KiB each); 125000 = the 1M-doc corpus of "
                         "BASELINE config 4 over 8 GPUs")
    ap.add_argument("--doc-len", type=int, default=1024)
    ap.add_argument("--nq", type=int, default=4, help="queries per step (32 tokens each); 4 = BASELINE config 1's query batch")
    ap.add_argument("--regimes", type=str, default=None,
                    help="other query batches measured after the headline and reported under 'regimes' ('' = none; default: REGIMES_SINGLE "
                         "at --gpus 1, the shorter REGIMES_MULTI above): N = N queries of "
                         "--q-len tokens; NxL = N queries of L tokens; NxrA-B = N queries of ragged lengths U{A..B} (real query lengths: "
                         "processing_utils.py:86 appends 10 augmentation tokens, SURVEY: Lq ~ 20-40)")
    ap.add_argument("--q-len", type=int, default=32)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    return ap.parse_args()


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_launch_command(n_gpus, argv, port, environ):
    """(command, environment) of the self-launch: `python -m torch.distributed.run`, one rank per GPU, rendezvous on 127.0.0.1 (the
    container's hostname may not resolve), HSA_ENABLE_IPC_MODE_LEGACY=0 (the host driver only supports dmabuf IPC: without it RCCL
    fails with `hipIpcGetMemHandle: invalid argument`), the host's threads divided between the ranks."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(environ, BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n_gpus)))
    return cmd, env


def launch_ranks(n_gpus):
    """`--gpus N` without a launcher: run N ranks of this file under torch.distributed.run (one per GPU, RCCL) and forward
    rank 0's JSON line.  Refuses (exit code 2) when fewer than N GPUs are visible, unless BENCH_SHARE_GPU=1."""
    import subprocess

    visible = torch.cuda.device_count()
    if visible < n_gpus and os.environ.get("BENCH_SHARE_GPU") != "1":
        sys.stderr.write(f"bench.py: --gpus {n_gpus} requested but only {visible} GPU(s) are visible; refusing to run "
                         f"{n_gpus} ranks on fewer devices (set BENCH_SHARE_GPU=1 for a plumbing-only run)\n")
        return 2
    cmd, env = rank_launch_command(n_gpus, sys.argv[1:], _free_port(), os.environ)
    return subprocess.call(cmd, env=env)


# N > 1: the regimes every rank runs after the headline.  Eight ranks each re-score their parity sample on the cores torchrun
# leaves them and the driver's clock covers the whole run, so the multi-GPU line carries the regimes that say something about
# scaling -- the HBM-bound headline's neighbours and config 4's 1000-query batch, uniform and ragged -- not the single-GPU ridge sweep.
REGIMES_SINGLE = "1,6,8,10,12,16,20,32,40,1000,4x40,16x40,1000x20,1000x40,1000x48,1000xr12-48"
REGIMES_MULTI = "1,32,1000,1000xr12-48"


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))
    # The contract is ONE JSON line on stdout.  Libraries print there too (RCCL writes its version banner to stdout when the first
    # communicator is created): keep the real stdout aside and point file descriptor 1 at stderr until the line is written.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.regimes is None:
        args.regimes = REGIMES_SINGLE if world == 1 else REGIMES_MULTI
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher must start one rank per GPU")
    # BENCH_SHARE_GPU=1 (+ BENCH_DIST_BACKEND=gloo): plumbing test of the N>1 path on a 1-GPU box (not a measurement)
    share_gpu = os.environ.get("BENCH_SHARE_GPU") == "1"
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("bench.py needs an MI355X: no GPU is visible to torch (there is no CPU fallback)")
    if world > n_dev and not share_gpu:
        raise SystemExit(f"{world} ranks but only {n_dev} visible GPU(s) (BENCH_SHARE_GPU=1 allows it for plumbing checks only)")
    dev_index = local_rank % n_dev if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist  # "nccl" is RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    import colpali_amd as amd
    from bench_legs import line as bench_line
    from bench_legs.resident import (forced_collective_numbers, mfma_ceiling, power_sample, ragged_docs_numbers, run_regime, short_docs_numbers,
                                     stream_ceiling, topk_parity, topk_vs_reference_fp32, wide_320_numbers)

    amd._lib.lib()  # fail loudly if the HIP library is missing
    # torch sizes its intra-op pool by the host's CPU count (128 threads on a 256-CPU box) -- not by what the container may use
    # (cgroup cpu.max: 16 CPUs per 100 ms on the GPU boxes).  128 OpenMP threads spinning behind any CPU-side torch op run that quota
    # dry and the kernel freezes the WHOLE process for the rest of the period: HIP-event kernel times of single launches came out
    # 5-10 x too long in this file's regimes (the launch sat between two event records while the host was frozen), and the drop-in
    # call stalled 70-90 ms in one call out of four (rounds 1-4; profiles/r05_logs/dropin_stalls.log).  Use what is granted.
    granted = amd._lib.effective_cpus()
    if torch.get_num_threads() > granted:
        torch.set_num_threads(granted)
    corpus = make_shard(args.docs, args.doc_len, dev, seed=1234 + rank)
    corpus.id_base = rank * args.docs
    q = make_queries(args.nq, args.q_len, dev, seed=99)
    torch.cuda.synchronize()

    dt, kern_ms, scores, top = run_regime(amd, q, corpus, args.steps, args.warmup, args.topk, world, rank, dist)
    kern_avg = sum(kern_ms) / len(kern_ms)
    pairs_per_step = args.nq * args.docs * world
    out = {
        "metric": "MaxSim (query,doc) pairs scored/sec",
        "value": pairs_per_step * args.steps / dt,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {
            "workload": f"{args.nq} query x {args.q_len} tokens vs resident pre-embedded shard of {args.docs} docs x "
                        f"{args.doc_len} patches x d=128 bf16 per GPU ({args.docs * args.doc_len * 256 / 2**30:.1f} GiB/GPU), "
                        f"fused MaxSim + per-shard top-{args.topk}" + ((" + RCCL all-gather merge" if os.environ.get("BENCH_DIST_BACKEND", "nccl") == "nccl" else " + gloo all-gather merge (plumbing run)") if world > 1 else ""),
            "docs_per_gpu": args.docs, "doc_len": args.doc_len, "n_queries": args.nq, "q_len": args.q_len,
            "top_k": args.topk, "parallelism": f"corpus-sharded x{world}",
        },
        "roofline": regime_numbers(args.nq, args.q_len, args.docs, args.doc_len, kern_avg),
    }
    if world > 1:
        # proof that the collective really spans `world` ranks, and every rank's own kernel time
        ones = torch.ones(1, dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        out["rccl_ranks"] = int(ones.item())
        out["dist_backend"] = "rccl (torch 'nccl')" if backend == "nccl" else backend
        mine = torch.tensor([kern_avg, float(dev_index)], dtype=torch.float64, device=ones.device)
        allk = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allk, mine)
        out["per_rank_kernel_ms"] = [float(t[0]) for t in allk]
        out["per_rank_device_index"] = [int(t[1]) for t in allk]
        out["gpus_visible_per_rank"] = n_dev
        out["shared_gpu_plumbing_run"] = bool(share_gpu)
        out["launched_by"] = "bench.py (self-spawned torch.distributed.run)" if os.environ.get("BENCH_SELF_LAUNCHED") else "external launcher"
        if out["rccl_ranks"] != args.gpus:
            raise SystemExit(f"process group spans {out['rccl_ranks']} ranks, --gpus {args.gpus} requested")
    ceil_ = stream_ceiling(amd, corpus) if corpus.blob.shape[1] == 128 else None
    if ceil_:
        out["roofline"]["stream_ceiling_gbs"] = ceil_["gbs"]
        out["roofline"]["stream_ceiling_what"] = ceil_["what"]
        if out["roofline"]["bound"] == "hbm":
            out["roofline"]["frac_of_stream_ceiling"] = out["roofline"]["achieved"] / ceil_["gbs"]
    ceil_m = mfma_ceiling(amd, corpus) if corpus.blob.shape[1] == 128 else None
    if ceil_m:
        out["mfma_ceiling"] = ceil_m
        if out["roofline"]["bound"] == "mfma":
            out["roofline"]["ratio_to_registers_only_probe"] = out["roofline"]["achieved"] / ceil_m["registers_only_tflops"]
    if not args.no_parity:
        # every rank checks its own shard (the CPU oracle as the checker); the verdicts are combined below
        local_top = amd.topk(scores, args.topk, corpus.id_base)
        n_rand = 1000 if world == 1 else 300       # N > 1: every rank runs the oracle on the cores torchrun leaves it (often one)
        par = topk_parity(amd, q, corpus, scores, local_top[0], local_top[1], args.topk, n_random=n_rand)
        par100 = topk_parity(amd, q, corpus, scores, *amd.topk(scores, 100, corpus.id_base), 100, n_queries=1, n_random=n_rand)
        par["k100"] = {k_: par100[k_] for k_ in ("checked_queries", "ids_equal", "ids_exact_equal", "max_rel_err")}
        if world == 1:
            try:    # the north star's own wording: ids against the reference einsum scorer's fp32 output, k = 10 and k = 100
                par["vs_reference_fp32"] = topk_vs_reference_fp32(amd, q, corpus, scores)
            except Exception as e:
                par["vs_reference_fp32"] = {"error": f"{type(e).__name__}: {e}"}
        if world > 1:
            flags = torch.tensor([int(par["ids_equal"]), int(par["ids_exact_equal"]), int(par100["ids_equal"]),
                                  int(par100["ids_exact_equal"])], dtype=torch.int32, device=ones.device)
            dist.all_reduce(flags, op=dist.ReduceOp.MIN)
            err = torch.tensor([max(par["max_rel_err"], par100["max_rel_err"])], dtype=torch.float64, device=ones.device)
            dist.all_reduce(err, op=dist.ReduceOp.MAX)
            par["ids_equal"], par["ids_exact_equal"] = bool(flags[0]), bool(flags[1])
            par["k100"]["ids_equal"], par["k100"]["ids_exact_equal"] = bool(flags[2]), bool(flags[3])
            par["max_rel_err"] = float(err.item())
            # the merge: the global list every rank holds must be the (score desc, id asc) top-k of the gathered local lists
            from oracle import topk_oracle

            msg = torch.cat([local_top[0].double().reshape(-1), local_top[1].double().reshape(-1)]).to(ones.device)
            allm = [torch.empty_like(msg) for _ in range(world)]
            dist.all_gather(allm, msg)
            nk = args.nq * args.topk
            cs = torch.stack([m[:nk].view(args.nq, args.topk) for m in allm], 1).reshape(args.nq, -1).float().cpu().numpy()
            ci = torch.stack([m[nk:].view(args.nq, args.topk) for m in allm], 1).reshape(args.nq, -1).long().cpu().numpy()
            ws, wi = topk_oracle.topk(cs, args.topk, ids=ci)
            ok = bool((wi == top[1].cpu().numpy()).all() and (ws == top[0].cpu().numpy()).all())
            okt = torch.tensor([int(ok)], dtype=torch.int32, device=ones.device)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            par["merge_equals_oracle_merge_on_all_ranks"] = bool(okt.item())
            par["ranks_checked"] = world
            # small corpora (plumbing runs): rank 0 also rebuilds EVERY shard from its seed, scores the unsharded corpus and takes its
            # top-k -- the list the collective returned must be that list, ids and scores (contiguous id ranges: shard r = ids r*docs ..)
            if args.docs * world <= 65536:
                same = True
                if rank == 0:
                    from colpali_amd.corpus import PackedCorpus

                    shards = [make_shard(args.docs, args.doc_len, dev, seed=1234 + r) for r in range(world)]
                    whole = PackedCorpus(blob=torch.cat([s.blob for s in shards]),
                                         offsets=(torch.arange(args.docs * world + 1, dtype=torch.int64) * args.doc_len).to(torch.int32).to(dev),
                                         clamp0=None, lengths=torch.full((args.docs * world,), args.doc_len, dtype=torch.int64))
                    ws_, wi_ = amd.topk(amd.maxsim_scores(q, whole), args.topk, 0)
                    same = bool(torch.equal(wi_, top[1]) and torch.equal(ws_, top[0]))
                    del shards, whole
                st_ = torch.tensor([int(same)], dtype=torch.int32, device=ones.device)
                dist.broadcast(st_, src=0)
                par["merged_topk_equals_unsharded_topk"] = bool(st_.item())
                par["id_base_per_rank"] = [r * args.docs for r in range(world)]
        if rank == 0:
            out["topk_parity"] = par
            out["parity_max_rel_err_vs_oracle_sample"] = par["max_rel_err"]
    if world == 1 and os.environ.get("BENCH_POWER_SAMPLE", "1") != "0":
        ps = power_sample(amd, q, corpus)
        if ps:
            out["roofline"]["power"] = ps
    if world == 1 and os.environ.get("BENCH_FORCE_COLLECTIVE", "1") != "0":
        out["forced_collective_1rank"] = forced_collective_numbers(amd, q, corpus, args.topk, dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from bench_legs.baselines import cpu_baseline, torch_gpu_reference
        from bench_legs.dropin import dropin_numbers, embed_and_score_numbers
        from bench_legs.head import embed_head_numbers

        out["cpu_baseline"] = cpu_baseline(args.q_len, args.doc_len)
        out["reference_on_this_gpu"] = torch_gpu_reference(args.q_len, args.doc_len)
        out["embed_head"] = embed_head_numbers(amd, dev)
        out["dropin_from_host_lists"] = dropin_numbers(amd)
        out["embed_and_score_1k_pages"] = embed_and_score_numbers(amd, dev)
        out["resident_colqwen2_page_geometry"] = ragged_docs_numbers(amd, dev, args.topk)
        out["resident_short_documents"] = short_docs_numbers(amd, dev)
        try:
            out["resident_width_320"] = wide_320_numbers(amd, dev)
        except Exception as e:
            out["resident_width_320"] = {"error": f"{type(e).__name__}: {e}"}

    # other regimes of the same step on the same resident shard (every rank takes part: collectives inside)
    regimes = []
    zero_corpus = None
    if world == 1 and os.environ.get("BENCH_ZERO_SHARD", "1") != "0":
        from colpali_amd.corpus import PackedCorpus

        zero_corpus = PackedCorpus(blob=torch.zeros_like(corpus.blob), offsets=corpus.offsets, clamp0=None, lengths=corpus.lengths)
    for spec in [x for x in args.regimes.split(",") if x]:
        nq, lens, len_label = parse_regime(spec, args.q_len)
        if "x" in spec:      # real query lengths: a host list of ragged / non-tile-sized queries, packed as the product packs them
            qq = amd.pack_queries(make_query_list(lens, seed=5 + nq + sum(lens)), dev)
        else:
            qq = make_queries(nq, args.q_len, dev, seed=5 + nq)
        steps = max(3, min(args.steps, 2000 // max(nq, 1)))
        d, km, _, _ = run_regime(amd, qq, corpus, steps, 2, args.topk, world, rank, dist)
        r = regime_numbers(nq, args.q_len, args.docs, args.doc_len, sum(km) / len(km), q_tokens=sum(lens))
        regimes.append({"n_queries": nq, "q_len": len_label, "q_tokens": sum(lens), "steps": steps,
                        "pairs_per_s": nq * args.docs * world * steps / d,
                        "pairs_per_s_per_32_real_tokens": nq * args.docs * world * steps / d * (sum(lens) / (32.0 * nq)),
                        "ms_per_step": d / steps * 1e3, "kernel_ms": r["kernel_ms"], "bound": r["bound"],
                        "frac": r["frac"], "hbm_gbs_per_gpu": r["hbm_gbs"], "mfma_tflops_per_gpu": r["mfma_tflops"],
                        "traffic": r["traffic"], "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"]})
        if ceil_m and r["bound"] == "mfma":
            regimes[-1]["ratio_to_registers_only_probe"] = r["mfma_tflops"] / ceil_m["registers_only_tflops"]
        if world == 1 and os.environ.get("BENCH_POWER_SAMPLE", "1") != "0":
            ps = power_sample(amd, qq, corpus)
            if ps:
                regimes[-1]["power"] = ps
        if zero_corpus is not None and nq <= 64:
            # the same launch on a zero-filled shard of the same shape: the same HBM traffic and instruction stream with operands that
            # toggle nothing, i.e. the kernel WITHOUT the socket's power cap -- splits "structure" from "cap" for every later reader
            _, kmz, _, _ = run_regime(amd, qq, zero_corpus, 3, 1, args.topk, world, rank, dist)
            rz = regime_numbers(nq, args.q_len, args.docs, args.doc_len, sum(kmz) / len(kmz), q_tokens=sum(lens))
            regimes[-1].setdefault("power", {})["frac_on_zeros"] = rz["frac"]
            regimes[-1]["power"]["kernel_ms_on_zeros"] = rz["kernel_ms"]
        del qq
    if zero_corpus is not None:
        _, kmz, _, _ = run_regime(amd, q, zero_corpus, 3, 1, args.topk, world, rank, dist)
        rz = regime_numbers(args.nq, args.q_len, args.docs, args.doc_len, sum(kmz) / len(kmz))
        out["roofline"].setdefault("power", {})["frac_on_zeros"] = rz["frac"]
        out["roofline"]["power"]["kernel_ms_on_zeros"] = rz["kernel_ms"]
        del zero_corpus
    out["regimes"] = regimes
    out["host_threads"] = {"torch_num_threads": torch.get_num_threads(), "cpus_the_container_grants": granted, "host_cpus": os.cpu_count()}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and os.environ.get("BENCH_LOSS", "1") != "0":
        try:
            from bench_legs.loss_step import loss_step_numbers

            out["loss_step_config5"] = loss_step_numbers(amd, dev)                 # BASELINE config 5 (after the timed regimes: its float64
        except Exception as e:                                                     # oracle is a minute of CPU work)
            out["loss_step_config5"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and os.environ.get("BENCH_VLM", "1") != "0":
        for key, family in (("embed_and_score_1k_pages_vlm_in_the_loop", "colpali"),              # BASELINE config 2
                            ("embed_and_score_1k_pages_vlm_in_the_loop_colqwen2", "colqwen2")):     # BASELINE config 3
            try:    # context only, and last: a multi-billion-parameter random-init VLM must never take the bench line down
                from bench_legs.vlm import vlm_in_the_loop_numbers

                out[key] = vlm_in_the_loop_numbers(amd, dev, family)
            except Exception as e:
                out[key] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        # the full report of every leg -> bench_detail.json (next to this file; also gpurun_out/ when it exists, which gpurun brings
        # back); stdout gets ONE compact strict-JSON line (bench_legs/line.py: <= 8 KiB -- round 5's 29 KB line went unparsed)
        detail_name = "bench_detail.json" if world == 1 else f"bench_detail_gpus{world}.json"
        written = None
        for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
            if os.path.isdir(d):
                try:
                    with open(os.path.join(d, detail_name), "w") as f:
                        f.write(bench_line.dumps_detail(out) + "\n")
                    written = written or detail_name
                except OSError:
                    pass
        sys.stdout.flush()
        os.write(real_stdout, (bench_line.dumps_line(out, written) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

