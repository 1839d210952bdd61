#!/usr/bin/env python
"""Width 320 (ColQwen3): the flat kernel K1bPF beside K1bP's query box -- ms per launch and the rate per REAL query token.

    python tools/ab_wide_flat.py                                   (shipped library: the dispatch's own choice)
    COLPALI_AMD_LIB=tools/_ab/libmaxsim_ab.so MSIM_PANELS_FLAT=0|1 python tools/ab_wide_flat.py      (force the box form / the flat form)

Cases (corpus: AB_DOCS pages of 1024 rows, unit rows): 1000 queries of 32 / 40 / 64 tokens as a box (msim_fwd picks), the same as a
flat token matrix (msim_fwd_ragged: always K1bPF), 1000 ragged queries U{12..48} flat and as the box padded to 48."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd
from colpali_amd.corpus import PackedCorpus

dev = torch.device("cuda:0")
docs, doc_len, dim = int(os.environ.get("AB_DOCS", "4096")), 1024, 320
dtype = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(1)
blob = torch.nn.functional.normalize(torch.randn((docs * doc_len, dim), generator=g, device=dev), dim=-1).to(dtype)
offsets = (torch.arange(docs + 1, dtype=torch.int64) * doc_len).to(torch.int32).to(dev)
corpus = PackedCorpus(blob=blob, offsets=offsets, clamp0=None, lengths=torch.full((docs,), doc_len, dtype=torch.int64))
print(f"library {os.environ.get('COLPALI_AMD_LIB', 'shipped')}  MSIM_PANELS_FLAT={os.environ.get('MSIM_PANELS_FLAT', '-')}  "
      f"corpus {docs} x {doc_len} x {dim} bf16 = {blob.numel() * 2 / 2**30:.2f} GiB", flush=True)


def timed(q, label, real_tokens):
    nq = len(q)
    out = torch.empty((nq, docs), dtype=torch.float32, device=dev)
    for _ in range(2):
        amd.maxsim_scores(q, corpus, out=out)
    reps = 5
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); amd.maxsim_scores(q, corpus, out=out); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[reps // 2]
    flops = 2.0 * real_tokens * docs * doc_len * dim
    print(f"{label:58s} {ms:9.3f} ms  {real_tokens * docs / ms / 1e6:8.2f} G(token x page)/s  {flops / ms / 1e9:7.1f} useful TF", flush=True)
    return out


nq = int(os.environ.get("AB_NQ", "1000"))
for lq in (32, 40, 64):
    box = torch.nn.functional.normalize(torch.randn((nq, lq, dim), generator=g, device=dev), dim=-1).to(dtype)
    a = timed(box, f"{nq} x Lq {lq}: box entry (msim_fwd)", nq * lq)
    b = timed(amd.pack_queries(box, dev, compact=False), f"{nq} x Lq {lq}: flat entry (msim_fwd_ragged, K1bPF)", nq * lq)
    print(f"    max |box - flat| = {float((a - b).abs().max()):.3e}", flush=True)
gl = torch.Generator().manual_seed(2)
lens = torch.randint(12, 49, (nq,), generator=gl).tolist()
tok = torch.nn.functional.normalize(torch.randn((sum(lens), dim), generator=g, device=dev), dim=-1).to(dtype)
qs = list(tok.split(lens))
flat = amd.pack_queries(qs, dev)
a = timed(flat, f"{nq} x U{{12..48}} ({sum(lens)} tokens): flat (K1bPF)", sum(lens))
box = torch.nn.utils.rnn.pad_sequence(qs, batch_first=True)
b = timed(box.contiguous(), f"{nq} x U{{12..48}} padded to {box.shape[1]}: box entry", sum(lens))
print(f"    max |flat - box| = {float((a - b).abs().max()):.3e}", flush=True)
