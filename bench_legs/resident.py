"""The resident-shard legs: the timed step (run_regime), the machine's own ceilings (tools/probe), power samples, the top-k parity check, the 1-rank RCCL merge, and the other corpus geometries (ColQwen2 pages, short documents, width 320)."""
from __future__ import annotations

import json
import os
import sys
import time

import torch

from .common import HBM_PEAK_GBS, MFMA_PEAK_TFLOPS, ROOT, make_queries, make_query_list, make_ragged_shard, make_shard, parse_regime, regime_numbers  # noqa: F401

def run_regime(amd, q, corpus, steps, warmup, topk, world, rank, dist):
    """Time `steps` full steps; returns (seconds for the K steps [max over ranks], kernel ms/launch list, the last
    step's score matrix, the last step's (top scores, top ids))."""
    dev = q.device
    scores = torch.empty((len(q), len(corpus)), dtype=torch.float32, device=dev)

    def step(ev=None):
        if ev is not None:
            ev[0].record()
        amd.maxsim_scores(q, corpus, out=scores)
        if ev is not None:
            ev[1].record()
        return amd.shard_topk(scores, topk, corpus.id_base, world, dist)

    top = None
    for _ in range(warmup):
        top = step()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        top = step(evs[i])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kern_ms = [a.elapsed_time(b) for a, b in evs]
    return dt, kern_ms, scores, top


def power_sample(amd, q, corpus, seconds=1.2):
    """Socket power and shader clock while msim_fwd runs back to back for `seconds` (rocm-smi sampled by a thread; context
    only).  The MI355X clocks to its power budget: next to a regime's roofline fraction this says whether the chip was at its
    cap (1400 W) and how much clock the power management took (2400 MHz nominal)."""
    import re
    import subprocess
    import threading

    smi = "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(smi):
        return None
    scores = torch.empty((len(q), len(corpus)), dtype=torch.float32, device=q.device)
    got, stop = [], threading.Event()

    def sampler():
        time.sleep(0.3)
        while not stop.is_set():
            try:
                out = subprocess.run([smi, "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
                card = json.loads(out)
                card = card[sorted(card.keys())[0]]
                pw = next((float(v) for k, v in card.items() if "ower" in k and re.match(r"^[0-9.]+$", str(v))), None)
                m = re.search(r"(\d+)\s*Mhz", next((str(v) for k, v in card.items() if k.lower().startswith("sclk")), ""), re.I)
                got.append((pw, int(m.group(1)) if m else None))
            except Exception:
                return

    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(max(1, 32 // len(q))):
            amd.maxsim_scores(q, corpus, out=scores)
        torch.cuda.synchronize()
    stop.set()
    th.join()
    pw = [p for p, _ in got if p is not None]
    ck = [c for _, c in got if c is not None]
    if not pw or not ck:
        return None
    return {"socket_power_w_avg": sum(pw) / len(pw), "socket_power_w_max": max(pw), "sclk_mhz_avg": sum(ck) / len(ck), "samples": len(got)}


def forced_collective_numbers(amd, q, corpus, topk, dev, steps=5):
    """The multi-GPU merge path on the ONE GPU this run has: a 1-rank `nccl` (= RCCL) process group, and the step of
    run_regime() with shard_topk(..., force_collective=True) -- message packing, all_gather_into_tensor on the uint8
    message, strided-view merge -- checked against the non-collective result.  Context only, never `value`."""
    import socket

    import torch.distributed as dist

    created = False
    try:
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
            created = True
        scores = torch.empty((q.shape[0], len(corpus)), dtype=torch.float32, device=dev)

        def step(force):
            amd.maxsim_scores(q, corpus, out=scores)
            return amd.shard_topk(scores, topk, corpus.id_base, 1, dist, force_collective=force)

        plain = step(False)
        forced = step(True)
        torch.cuda.synchronize()
        same = bool(torch.equal(plain[0], forced[0]) and torch.equal(plain[1], forced[1]))
        times = {}
        for force in (False, True):
            step(force)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step(force)
            torch.cuda.synchronize()
            times[force] = (time.perf_counter() - t0) / steps * 1e3
        return {"what": "1-rank nccl (RCCL) group on this GPU: per-shard top-k written into the 12 B/candidate message, "
                        "all_gather_into_tensor, merge on strided views of the gathered bytes",
                "backend": dist.get_backend(), "world": dist.get_world_size(), "ids_and_scores_equal_to_non_collective": same,
                "ms_per_step_non_collective": times[False], "ms_per_step_forced_collective": times[True]}
    except Exception as e:  # context only: never take the bench line down
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        if created:
            dist.destroy_process_group()


def stream_ceiling(amd, corpus):
    """The machine's own ceiling for K1s' document stream: the same LDS-DMA loads of the same resident shard with no MFMA, no
    max/sum and no output (msim_probe_stream, include/maxsim_probe.h: tools/probe/libmaxsim_probe.so, not the product library).  GB/s of the shard bytes; HIP events on the launch stream."""
    from tools import probe

    L = probe.lib()
    if L is None:
        return None
    rows = int(corpus.blob.shape[0]) // 256 * 256
    sink = torch.zeros(4, dtype=torch.float32, device=corpus.blob.device)
    st = torch.cuda.current_stream()
    ms = []
    for i in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        rc = L.msim_probe_stream(0, corpus.blob.data_ptr(), rows, 128, sink.data_ptr(), st.cuda_stream)
        b.record(st)
        torch.cuda.synchronize()
        if rc != 0:
            raise RuntimeError(f"msim_probe_stream failed: {L.msim_probe_last_error().decode()}")
        if i >= 2:
            ms.append(a.elapsed_time(b))
    t = sorted(ms)[len(ms) // 2]
    return {"gbs": rows * 256 / t / 1e6, "ms": t, "what": "msim_probe_stream(MSIM_PROBE_ROWS256B) over the same shard: "
            "K1s' loads without its arithmetic"}


def topk_parity(amd, q, corpus, scores, top_s, top_i, k, n_queries=2, n_random=1000):
    """SURVEY 8(d) C4-ii on this rank's shard: for `n_queries` sampled queries re-score the returned per-shard top-k plus
    `n_random` random documents with the CPU oracle (truth tier: fp32 inputs, double accumulate) and compare rankings.

    `ids_equal`: at every rank r the oracle score of the returned id equals the r-th best oracle score of the candidate set
    within twice the measured score error (a different id is only accepted between documents the two computations cannot
    tell apart); `ids_exact_equal`: the id lists are identical to the oracle's (score desc, id asc) ranking."""
    import numpy as np

    from oracle import maxsim_oracle as mo
    from oracle import topk_oracle

    n = len(corpus)
    gq = torch.Generator().manual_seed(17)
    qsel = torch.randperm(q.shape[0], generator=gq)[:n_queries].tolist()
    off = corpus.offsets.cpu().numpy().astype(np.int64)
    ids_equal, exact, max_err, n_cand = True, True, 0.0, 0
    for qi in qsel:
        ret = top_i[qi].cpu().numpy()
        ret_local = ret[ret >= 0] - corpus.id_base
        rnd = torch.randperm(n, generator=gq)[:n_random].numpy()
        cand = np.unique(np.concatenate([ret_local, rnd]))            # sorted local ids
        docs = [corpus.blob[int(off[c]):int(off[c + 1])].float().cpu().numpy() for c in cand]
        want = mo.score_multi_vector([q[qi].float().cpu().numpy()], docs, batch_size=10**9, mode="f32")[0]
        got = scores[qi, torch.from_numpy(cand).to(scores.device)].float().cpu().numpy()
        err = float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1.0)))
        max_err = max(max_err, err)
        kk = int(min(k, len(ret_local)))
        # the returned list must be the top-k of the candidate set (every returned id is in it; a random non-returned
        # document that out-scores a returned one is a ranking error)
        _, w_ids = topk_oracle.topk(want[None, :], kk, ids=(cand + corpus.id_base)[None, :])
        exact = exact and bool(np.array_equal(w_ids[0], ret[:kk]))
        order = np.argsort(-want, kind="stable")[:kk]
        pos = np.searchsorted(cand, ret_local[:kk])
        tol = 2.0 * err + 1e-7
        ids_equal = ids_equal and bool(np.all(np.abs(want[pos] - want[order]) <= tol * np.maximum(np.abs(want[order]), 1.0)))
        n_cand += len(cand)
    return {"checked_queries": len(qsel), "candidates_rescored": n_cand, "k": k, "ids_equal": ids_equal,
            "ids_exact_equal": exact, "max_rel_err": max_err,
            "what": "returned top-k + random docs of this rank's shard re-scored by the CPU oracle (fp32 inputs, double "
                    "accumulate); ids_equal tolerates swaps only between docs closer than 2 x max_rel_err"}


def topk_vs_reference_fp32(amd, q, corpus, scores, n_queries=2, n_random=1000):
    """North star: "bit-exact top-k doc indices" against the reference torch.einsum scorer.  For `n_queries` sampled queries: the ids the
    selection kernel returns at k = 10 and k = 100 over the WHOLE shard against the (score desc, id asc) ranking of the REFERENCE's own fp32
    scores (oracle/torch_port.py: processing_utils.py:170-186 with its torch calls, fp32 upcasts of the same bf16 rows) over the candidate
    set {our top-100} + `n_random` random documents -- every document that could enter the list is in the set.  Where the lists differ,
    `max_swap_gap_ulps` is the distance IN THE REFERENCE'S OWN SCORES between the two ids at that position, in fp32 ulps: two fp32
    contractions that sum in a different order cannot agree below that, and `reference_cpu_vs_reference_gpu_*` shows the reference
    against itself (its einsum on the host cores against its einsum on this GPU) doing the same."""
    import numpy as np

    from oracle import topk_oracle, torch_port

    n = len(corpus)
    gq = torch.Generator().manual_seed(23)
    qsel = torch.randperm(q.shape[0], generator=gq)[:n_queries].tolist()
    off = corpus.offsets.cpu().numpy().astype(np.int64)
    out = {"checked_queries": len(qsel), "k10_ids_exact_equal": True, "k100_ids_exact_equal": True, "k10_differing_positions": 0,
           "k100_differing_positions": 0, "k100_max_swap_gap_ulps": 0.0, "reference_cpu_vs_reference_gpu_k100_ids_equal": True,
           "reference_cpu_vs_reference_gpu_k100_differing_positions": 0, "log": []}
    for qi in qsel:
        _, ids100 = amd.topk(scores[qi:qi + 1], 100, corpus.id_base)
        ours = ids100[0].cpu().numpy()
        ours = ours[ours >= 0]
        rnd = torch.randperm(n, generator=gq)[:n_random].numpy()
        cand = np.unique(np.concatenate([ours - corpus.id_base, rnd]))
        docs = [corpus.blob[int(off[c]):int(off[c + 1])].float().cpu() for c in cand]
        qq = [q[qi].float().cpu()]
        ref_cpu = torch_port.score_multi_vector_cpu(qq, docs, device="cpu")[0].numpy()
        ref_gpu = torch_port.score_multi_vector_cpu(qq, docs, device=str(corpus.blob.device))[0].numpy()
        ids = (cand + corpus.id_base)[None, :]
        for k in (10, 100):
            kk = min(k, len(ours))
            _, want = topk_oracle.topk(ref_cpu[None, :], kk, ids=ids)
            diff = np.nonzero(want[0] != ours[:kk])[0]
            out[f"k{k}_ids_exact_equal"] = out[f"k{k}_ids_exact_equal"] and len(diff) == 0
            out[f"k{k}_differing_positions"] += int(len(diff))
            for r in diff:
                a, b = int(ours[r]), int(want[0][r])
                sa, sb = (float(ref_cpu[np.searchsorted(cand, x - corpus.id_base)]) for x in (a, b))
                ulps = abs(sa - sb) / float(np.spacing(np.float32(max(abs(sa), abs(sb)))))
                out["k100_max_swap_gap_ulps"] = max(out["k100_max_swap_gap_ulps"], ulps)
                if len(out["log"]) < 24:
                    out["log"].append({"query": qi, "k": k, "rank": int(r), "ours": a, "reference": b, "reference_score_of_ours": sa,
                                       "reference_score_of_its_own": sb, "gap_in_fp32_ulps": ulps})
        _, w_cpu = topk_oracle.topk(ref_cpu[None, :], min(100, len(ours)), ids=ids)
        _, w_gpu = topk_oracle.topk(ref_gpu[None, :], min(100, len(ours)), ids=ids)
        d = int((w_cpu != w_gpu).sum())
        out["reference_cpu_vs_reference_gpu_k100_ids_equal"] = out["reference_cpu_vs_reference_gpu_k100_ids_equal"] and d == 0
        out["reference_cpu_vs_reference_gpu_k100_differing_positions"] += d
    out["what"] = ("our top-10 / top-100 ids over the whole shard vs the (score desc, id asc) ranking of the reference's fp32 einsum scores "
                   "(host cores) over {our top-100} + random documents; swap gaps measured in the reference's own scores")
    return out


def mfma_ceiling(amd, corpus):
    """The machine's own matrix-core ceiling under its power budget (msim_probe_mfma, include/maxsim_probe.h: tools/probe/libmaxsim_probe.so, not the product library): back-to-back
    v_mfma_f32_16x16x32_bf16 on rows of the resident shard (the operand values the scorer multiplies), two waves per SIMD, no HBM
    traffic.  `kernel_mix` = with K1b's operand path (A fragments re-read from LDS) and its max folds; `registers_only` = nothing
    but MFMAs.  MI355X clocks to its power budget: on real operand values the chip does not reach the 2.5 PFLOP/s of
    1024 SIMDs x 1024 FLOP/clk x 2.4 GHz (on zeros it nearly does), so this is what an MFMA-bound kernel can be held against."""
    from tools import probe

    L = probe.lib()
    rows = int(corpus.blob.shape[0])
    if L is None or rows < 256 * 8 * 5 * 32:
        return None
    sink = torch.zeros(4, dtype=torch.float32, device=corpus.blob.device)
    st = torch.cuda.current_stream()
    iters = 4000
    flop = 256 * 8 * iters * 32 * 32768
    out = {}
    for name, variant in (("kernel_mix", 7), ("registers_only", 4), ("kernel_mix_32x32x16_tiles", 3)):
        ms = []
        for i in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            rc = L.msim_probe_mfma(variant, corpus.blob.data_ptr(), rows, iters, sink.data_ptr(), st.cuda_stream)
            b.record(st)
            torch.cuda.synchronize()
            if rc != 0:
                raise RuntimeError(f"msim_probe_mfma failed: {L.msim_probe_last_error().decode()}")
            if i >= 2:
                ms.append(a.elapsed_time(b))
        out[name + "_tflops"] = flop / sorted(ms)[len(ms) // 2] / 1e9
    out["what"] = ("msim_probe_mfma on rows of the resident shard: v_mfma_f32_16x16x32_bf16 (the scorers' tile shape) back to back, 2 waves "
                   "per SIMD, no memory traffic; kernel_mix = A fragments from LDS + max folds (K1s / K1b's instruction mix), "
                   "registers_only = MFMAs alone; kernel_mix_32x32x16_tiles = the same mix on the 32x32x16 tile the kernels used before")
    return out


def ragged_docs_numbers(amd, dev, topk):
    """The resident path on BASELINE config 3's page geometry (ColQwen2: 267..779 patch rows per page, here 60 000 pages = 8 GiB): the
    HBM-bound and the MFMA-bound regime with ragged documents AND ragged queries.  Context (the headline shard is config 4's 1024-row pages)."""
    corpus = make_ragged_shard(60000, 267, 779, dev, seed=77)
    rows = int(corpus.blob.shape[0])
    out = {"workload": f"60000 pages x U{{267..779}} rows ({rows} rows, {rows * 256 / 2**30:.1f} GiB resident)"}
    for name, lens in (("4_queries_x_32", [32] * 4), ("4_queries_ragged_12_48", parse_regime("4xr12-48", 32)[1]),
                       ("1000_queries_ragged_12_48", parse_regime("1000xr12-48", 32)[1])):
        q = amd.pack_queries(make_query_list(lens, seed=sum(lens)), dev)
        scores = torch.empty((len(lens), len(corpus)), dtype=torch.float32, device=dev)
        for _ in range(2):
            amd.maxsim_scores(q, corpus, out=scores)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in evs:
            a.record(); amd.maxsim_scores(q, corpus, out=scores); b.record()
        amd.topk(scores, topk)
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)[2]
        alg = rows * 256 + sum(lens) * 256 + len(lens) * len(corpus) * 4
        flops = 2.0 * sum(lens) * rows * 128
        gbs, tf = alg / ms / 1e6, flops / ms / 1e9
        bound = "hbm" if alg / HBM_PEAK_GBS / 1e9 >= flops / MFMA_PEAK_TFLOPS / 1e12 else "mfma"
        out[name] = {"kernel_ms": ms, "pairs_per_s": len(lens) * len(corpus) / ms * 1e3, "hbm_gbs": gbs, "mfma_tflops": tf, "bound": bound,
                     "frac": gbs / HBM_PEAK_GBS if bound == "hbm" else tf / MFMA_PEAK_TFLOPS, "q_tokens": sum(lens)}
    del corpus
    return out


def short_docs_numbers(amd, dev):
    """The resident path on SHORT documents (round-4 review, weak 8): a token-pooled corpus -- pool factor 3 of a 1030-patch page
    (README.md:225, compression/token_pooling) = 343 rows -- and 64-row documents, 8 GiB of rows each, in the HBM-bound and the
    MFMA-bound regime.  K1b pays one chunk barrier, one table write and one pass of token sums per document: the numbers show what
    that costs (the structural fix -- several documents per chunk -- is not built, DESIGN.md section 8, gap 4)."""
    out = {}
    for name, doc_len in (("pooled_343_rows", 343), ("64_rows", 64)):
        n_docs = (8 << 30) // (doc_len * 256)
        corpus = make_shard(n_docs, doc_len, dev, seed=5)
        leg = {"docs": n_docs, "doc_len": doc_len}
        for qname, lens in (("4_queries_x_32", [32] * 4), ("1000_queries_x_32", [32] * 1000), ("1000_queries_ragged_12_48", parse_regime("1000xr12-48", 32)[1])):
            q = amd.pack_queries(make_query_list(lens, seed=sum(lens) + doc_len), dev)
            scores = torch.empty((len(lens), n_docs), dtype=torch.float32, device=dev)
            amd.maxsim_scores(q, corpus, out=scores)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
            for a, b in evs:
                a.record(); amd.maxsim_scores(q, corpus, out=scores); b.record()
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in evs)[1]
            r = regime_numbers(len(lens), 32, n_docs, doc_len, ms, q_tokens=sum(lens))
            leg[qname] = {"kernel_ms": ms, "bound": r["bound"], "frac": r["frac"], "hbm_gbs": r["hbm_gbs"], "mfma_tflops": r["mfma_tflops"]}
            del scores
        out[name] = leg
        del corpus
    return out


def wide_320_numbers(amd, dev):
    """Width 320 (ColQwen3, models/qwen3/colqwen3/modeling_colqwen3.py:48) on the panel kernels: 4 096 pages x 1 024 rows x 320 bf16
    (2.5 GiB resident).  4 queries: K1sP (HBM-bound); 1000 x 32: K1bP's query box; 1000 x 40 and 1000 ragged U{12..48}: the flat
    kernel K1bPF (round 5), whose rate per REAL token should sit within 10 % of the Lq 32 one."""
    n_docs, doc_len, dim = 4096, 1024, 320
    g = torch.Generator(device=dev).manual_seed(11)
    blob = torch.nn.functional.normalize(torch.randn((n_docs * doc_len, dim), generator=g, device=dev), dim=-1).to(torch.bfloat16)
    from colpali_amd.corpus import PackedCorpus
    corpus = PackedCorpus(blob=blob, offsets=(torch.arange(n_docs + 1, dtype=torch.int64) * doc_len).to(torch.int32).to(dev), clamp0=None,
                          lengths=torch.full((n_docs,), doc_len, dtype=torch.int64))
    out = {"workload": f"{n_docs} pages x {doc_len} rows x {dim} bf16 ({blob.numel() * 2 / 2**30:.1f} GiB resident)"}
    for name, lens in (("4_queries_x_32", [32] * 4), ("1000_queries_x_32", [32] * 1000), ("1000_queries_x_40", [40] * 1000),
                       ("1000_queries_ragged_12_48", parse_regime("1000xr12-48", 32)[1])):
        tok = torch.nn.functional.normalize(torch.randn((sum(lens), dim), generator=g, device=dev), dim=-1).to(torch.bfloat16)
        uniform = len(set(lens)) == 1
        q = tok.view(len(lens), lens[0], dim) if uniform else amd.pack_queries(list(tok.split(lens)), dev)     # a box (msim_fwd picks) / flat
        scores = torch.empty((len(lens), n_docs), dtype=torch.float32, device=dev)
        amd.maxsim_scores(q, corpus, out=scores)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
        for a, b in evs:
            a.record(); amd.maxsim_scores(q, corpus, out=scores); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)[1]
        alg = blob.numel() * 2 + sum(lens) * dim * 2 + len(lens) * n_docs * 4
        flops = 2.0 * sum(lens) * n_docs * doc_len * dim
        gbs, tf = alg / ms / 1e6, flops / ms / 1e9
        bound = "hbm" if alg / HBM_PEAK_GBS / 1e9 >= flops / MFMA_PEAK_TFLOPS / 1e12 else "mfma"
        out[name] = {"kernel_ms": ms, "q_tokens": sum(lens), "real_token_pages_per_s": sum(lens) * n_docs / ms * 1e3, "hbm_gbs": gbs,
                     "useful_mfma_tflops": tf, "bound": bound, "frac": gbs / HBM_PEAK_GBS if bound == "hbm" else tf / MFMA_PEAK_TFLOPS}
        del scores
    out["ragged_per_real_token_rate_vs_Lq32"] = (out["1000_queries_ragged_12_48"]["real_token_pages_per_s"] /
                                                 out["1000_queries_x_32"]["real_token_pages_per_s"])
    del corpus, blob
    return out
