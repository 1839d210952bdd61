"""BASELINE config 5: one rank's loss step (forward + backward), eager and as one hipGraph, against the reference module and the float64 oracle."""
from __future__ import annotations

import json
import os
import sys
import time

import torch

from .common import HBM_PEAK_GBS, MFMA_PEAK_TFLOPS, ROOT, make_queries, make_query_list, make_ragged_shard, make_shard, parse_regime, regime_numbers  # noqa: F401

def _median_ms(fn, reps, sync=True):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


def _graph_of(step, warm=3):
    """`step` captured once as a hipGraph (torch.cuda.graph) after `warm` eager runs on a side stream."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warm):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    graph.replay()
    torch.cuda.synchronize()
    return graph


def _graph_device_ms(graph, replays=50):
    """Device time of one replay: HIP events on the replay stream around `replays` back-to-back replays (no host gap between them)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(replays):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / replays


def loss_step_numbers(amd, dev):
    """BASELINE config 5 -- "ColbertPairwiseCELoss training step, in-batch negatives bs=256, ColQwen2, 8 x MI355X data-parallel" -- as ONE
    rank sees it: B = 32 local queries [32, Lq, 128] against the C = 256 pages gathered from the 8 ranks [256, 780, 128], offset = rank * B
    (trainer/contrastive_trainer.py:135-162), and the trainer's symmetric direction (:202-206: the local pages as `query_embeddings`
    [32, 780, 128] against the gathered queries [256, Lq, 128]); forward + backward of the loss alone (the three VLM forwards around it
    stay on PyTorch-ROCm), bf16.  Per loss class: eager and one-hipGraph time of the forward direction and of BOTH directions captured as
    one graph, device time of a replay (HIP events), peak MiB, the fraction of the roof
        max(2 * B * C * Lq * Ld * 128 FLOP / 2.5 PFLOP/s, (Q + D read, dQ + dD written) / 8 TB/s)     per direction,
    the reference's own module (the verbatim late_interaction_losses.py where the fetched copy exists, else its restatement) on the same
    GPU, and loss / gradient error against the float64 oracle (oracle/li_loss_oracle.py: the checker, not the thing measured)."""
    from oracle import li_loss_oracle, refimport

    B, C, Ld, off = 32, 256, 780, 96
    g = torch.Generator(device=dev).manual_seed(55)

    def unit(*shape):
        return torch.nn.functional.normalize(torch.randn(shape, generator=g, device=dev), dim=-1).to(torch.bfloat16)

    try:
        _, ref_mod, ref_kind = refimport.load_hot_path()
    except Exception:
        ref_mod, ref_kind = None, "port"

    def ref_port(kind):
        def f(q, d, offset=0):            # late_interaction_losses.py:296-313 / :152-164 restated with the reference's own torch calls
            lengths = (q[:, :, 0] != 0).sum(dim=1)
            scores = torch.einsum("bnd,csd->bcns", q, d).amax(dim=3).sum(dim=2) / lengths.unsqueeze(1)
            if kind == "pairwise":
                pos = scores.diagonal(offset=offset)
                top2 = scores.topk(2, dim=1).values
                neg = torch.where(top2[:, 0] == pos, top2[:, 1], top2[:, 0])
                return torch.nn.functional.softplus(neg - pos).mean()
            return torch.nn.functional.cross_entropy(scores / 0.02, torch.arange(q.shape[0], device=q.device) + offset)
        return f

    out = {"shape": {"B": B, "C": C, "Ld": Ld, "dim": 128, "offset": off, "dtype": "bf16"}, "reference_module": ref_kind}
    for qname, lens in (("Lq32", [32] * C), ("Lq_ragged_20-40_left_padded_to_40", torch.randint(20, 41, (C,), generator=torch.Generator().manual_seed(3)).tolist())):
        Lq = max(lens)
        Qg = unit(C, Lq, 128)                                     # the queries of all ranks (the symmetric direction's gathered side)
        for c, n in enumerate(lens):
            Qg[c, : Lq - n] = 0                                   # left padding: rows exactly zero (modeling_colqwen2.py:36, :69)
        D = unit(C, Ld, 128)                                      # the pages of all ranks
        for b in range(B):                                        # positives: noisy copies of the query's tokens somewhere in its page
            rows = torch.randperm(Ld, generator=torch.Generator().manual_seed(b))[:Lq].to(dev)
            D[off + b, rows] = torch.nn.functional.normalize(Qg[off + b].float() + 0.6 * torch.randn((Lq, 128), generator=g, device=dev), dim=-1).to(torch.bfloat16) * (Qg[off + b, :, :1] != 0)
        Q, P = Qg[off:off + B].clone(), D[off:off + B].clone()    # this rank's own queries and pages
        flop = 2.0 * B * C * sum(lens[off:off + B]) / B * Ld * 128          # forward direction, real tokens only
        flop_sym = 2.0 * B * Ld * sum(lens) * 128
        bytes_dir = 2.0 * 2 * (B * Lq + C * Ld) * 128                        # Q, D read + dQ, dD written, bf16
        roof_fwd_ms = max(flop / (MFMA_PEAK_TFLOPS * 1e12), bytes_dir / (HBM_PEAK_GBS * 1e9)) * 1e3
        roof_sym_ms = max(flop_sym / (MFMA_PEAK_TFLOPS * 1e12), 2.0 * 2 * (B * Ld + C * Lq) * 128 / (HBM_PEAK_GBS * 1e9)) * 1e3
        legs = {}
        for cls, kind in (("ColbertPairwiseCELoss", "pairwise"), ("ColbertLoss", "infonce")):
            ours = getattr(amd, cls)()
            refs = getattr(ref_mod, cls)() if ref_mod is not None else ref_port(kind)
            leaves = [t.detach().clone().requires_grad_(True) for t in (Q, D, P, Qg)]

            def one(mod, q=leaves[0], d=leaves[1]):
                q.grad = d.grad = None
                mod(query_embeddings=q, doc_embeddings=d, offset=off).backward()

            def both(mod, ls=leaves):
                for t in ls:
                    t.grad = None
                l1 = mod(query_embeddings=ls[0], doc_embeddings=ls[1], offset=off)      # contrastive_trainer.py:198
                l2 = mod(query_embeddings=ls[2], doc_embeddings=ls[3], offset=off)      # :202-206, compute_symetric_loss
                ((l1 + l2) / 2).backward()

            r = {}
            for tag, mod in (("ours", ours), ("reference_on_this_gpu", refs)):
                rr = {}
                for sname, step in (("forward_direction", lambda m=mod: one(m)), ("both_directions", lambda m=mod: both(m))):
                    for _ in range(3):
                        step()
                    for t in leaves:                 # what the step itself allocates at its peak, gradients included: nothing of a
                        t.grad = None                # previous step alive when the baseline is taken
                    torch.cuda.synchronize()
                    torch.cuda.reset_peak_memory_stats()
                    base = torch.cuda.memory_allocated()
                    step()
                    torch.cuda.synchronize()
                    rr[sname] = {"eager_ms": _median_ms(step, 11), "peak_mib": (torch.cuda.max_memory_allocated() - base) / 2**20}
                    if tag == "ours":
                        graph = _graph_of(step)
                        rr[sname]["one_hipgraph_ms"] = _median_ms(graph.replay, 21)
                        rr[sname]["one_hipgraph_device_ms"] = _graph_device_ms(graph)
                        roof = roof_fwd_ms if sname == "forward_direction" else roof_fwd_ms + roof_sym_ms
                        rr[sname]["roof_ms"] = roof
                        rr[sname]["frac_of_roof"] = roof / rr[sname]["one_hipgraph_device_ms"]
                        del graph
                r[tag] = rr
            # parity of what was just timed: loss and gradients of the forward direction against the float64 oracle (CPU, the checker)
            one(ours)
            torch.cuda.synchronize()
            want_loss, want_dq, want_dd = li_loss_oracle.loss_and_grads(kind, Q.float().cpu(), D.float().cpu(), offset=off)
            got_loss = float(ours(query_embeddings=leaves[0], doc_embeddings=leaves[1], offset=off).detach().float())
            # padding rows (exactly zero) are excluded: every similarity of such a row ties at 0, the reference's amax backward splits
            # the gradient evenly, ours routes it to the first row, and the model multiplies it by the attention mask either way
            q_real = (Q.float().abs().sum(-1, keepdim=True) > 0).cpu()
            rel = lambda got, want, m=None: float(((got.detach().double().cpu() - want) * (1 if m is None else m)).abs().max() / want.abs().max().clamp_min(1e-30))   # noqa: E731
            r["parity_vs_float64_oracle"] = {"loss": got_loss, "loss_oracle": float(want_loss),
                                             "loss_rel_err": abs(got_loss - float(want_loss)) / max(abs(float(want_loss)), 1e-30),
                                             "dQ_max_err_over_max_abs": rel(leaves[0].grad, want_dq, q_real),
                                             "dD_max_err_over_max_abs": rel(leaves[1].grad, want_dd),
                                             "note": "bf16 loss / gradients (one rounding of an fp32 result) against float64 on the same bf16-valued inputs; "
                                                     "zero (padding) query rows excluded from dQ"}
            r["speedup_vs_reference_both_directions_eager"] = r["reference_on_this_gpu"]["both_directions"]["eager_ms"] / r["ours"]["both_directions"]["eager_ms"]
            legs[cls] = r
            del leaves
        out[qname] = legs
    return out
