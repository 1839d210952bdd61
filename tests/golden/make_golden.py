"""Generate the golden fixtures under tests/golden/ from the LIVE reference.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py

Every fixture stores the exact inputs (bf16 as uint16 bit patterns, or fp32)
and the outputs the reference functions returned for them:
  colpali_engine/utils/processing_utils.py:132-187  score_multi_vector
  colpali_engine/loss/late_interaction_losses.py:255-313  ColbertPairwiseCELoss
(+ ColbertLoss :110-164 on the same inputs, for the shared MaxSim core).
The fixtures are what pins oracle/ and, through it, the HIP path.
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle import refimport  # noqa: E402

P, L = refimport.load()
torch.set_num_threads(8)


def unit_rows(n, dim, g, dtype=torch.bfloat16):
    return F.normalize(torch.randn(n, dim, generator=g), dim=-1).to(dtype)


def bits(t: torch.Tensor) -> np.ndarray:
    assert t.dtype == torch.bfloat16
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def ref_score(qs, ps, batch_size=128):
    """(truth = reference on fp32 upcasts, literal = reference on the raw tensors)."""
    truth = P.score_multi_vector([q.float() for q in qs], [p.float() for p in ps],
                                 batch_size=batch_size, device="cpu")
    literal = P.score_multi_vector(qs, ps, batch_size=batch_size, device="cpu")
    return truth.numpy(), literal.numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def ragged_case(seed, q_lens, p_lens, dim, batch_sizes, name):
    g = torch.Generator().manual_seed(seed)
    qs = [unit_rows(n, dim, g) for n in q_lens]
    ps = [unit_rows(n, dim, g) for n in p_lens]
    out = dict(
        q_lens=np.array(q_lens, np.int32), p_lens=np.array(p_lens, np.int32), dim=np.int32(dim),
        q_bits=np.concatenate([bits(q) for q in qs]), p_bits=np.concatenate([bits(p) for p in ps]),
        batch_sizes=np.array(batch_sizes, np.int32),
    )
    for bs in batch_sizes:
        truth, literal = ref_score(qs, ps, bs)
        out[f"truth_bs{bs}"] = truth
        out[f"literal_bs{bs}"] = literal
    save(name, **out)


def ragged_d320():
    # (1b) the same at ColQwen3's width (models/qwen3/colqwen3/modeling_colqwen3.py:48): ragged queries straddling 16-token units and
    #      32-token tiles -- what the flat panel kernel K1bPF sees (round 5)
    ragged_case(13, [5, 40, 17, 33, 48, 12, 1, 20, 64], [64, 33, 100, 1, 47, 300, 128, 31, 32, 160, 7], 320, [128, 4, 1],
                "score_ragged_d320.npz")


def main():
    # (1) small ragged lists, d=128: exercises block padding (finding 4), tails, 1-row docs
    ragged_case(11, [5, 32, 17, 1], [64, 33, 100, 1, 47, 96, 128, 31, 32, 160], 128, [128, 4, 3, 1],
                "score_ragged_d128.npz")
    ragged_d320()

    # (2) config 1 of BASELINE.json: 4 queries x 16 docs, [32,128] x [1024,128] bf16.
    #     Inputs are regenerated from the seed by the tests; the sha256 pins them.
    g = torch.Generator().manual_seed(1234)
    qs = [unit_rows(32, 128, g) for _ in range(4)]
    ps = [unit_rows(1024, 128, g) for _ in range(16)]
    truth, literal = ref_score(qs, ps)
    h = hashlib.sha256()
    for t in qs + ps:
        h.update(bits(t).tobytes())
    save("score_config1.npz", seed=np.int64(1234), n_q=np.int32(4), n_d=np.int32(16), Lq=np.int32(32),
         Ld=np.int32(1024), dim=np.int32(128), sha256=np.frombuffer(h.digest(), dtype=np.uint8),
         truth=truth, literal=literal)

    # (3) negative-similarity documents: alone vs sharing a block with a longer one (finding 4)
    g = torch.Generator().manual_seed(7)
    q = unit_rows(8, 128, g)
    anti = -q.float().sum(0, keepdim=True)                    # negative similarity with EVERY query token
    short = F.normalize(anti + 0.02 * torch.randn(3, 128, generator=g), dim=-1).to(torch.bfloat16)
    assert (q.float() @ short.float().T).max() < 0
    long_ = unit_rows(40, 128, g)
    t_alone, l_alone = ref_score([q], [short])
    t_block, l_block = ref_score([q], [short, long_])
    t_split, l_split = ref_score([q], [short, long_], batch_size=1)
    save("score_negative_clamp.npz", q_bits=bits(q), short_bits=bits(short), long_bits=bits(long_),
         truth_alone=t_alone, truth_block=t_block, truth_split=t_split,
         literal_alone=l_alone, literal_block=l_block, literal_split=l_split)

    # (4) 3-D tensor inputs with physical zero rows (what the models emit: proj * attention_mask)
    g = torch.Generator().manual_seed(21)
    q3 = torch.stack([unit_rows(12, 128, g) for _ in range(3)])
    q3[1, 9:] = 0
    p3 = torch.stack([unit_rows(70, 128, g) for _ in range(5)])
    p3[0, :20] = 0          # left padded (ColQwen2 style)
    p3[3, 50:] = 0          # right padded (ColPali style)
    truth = P.score_multi_vector(q3.float(), p3.float(), device="cpu").numpy()
    literal = P.score_multi_vector(q3, p3, device="cpu").numpy()
    save("score_tensor3d.npz", q_bits=bits(q3), p_bits=bits(p3), q_shape=np.array(q3.shape, np.int32),
         p_shape=np.array(p3.shape, np.int32), truth=truth, literal=literal)

    # (5) the reference's own unit-test shape: d=32 fp32 lists (tests/utils/test_processing_utils.py:15-35)
    g = torch.Generator().manual_seed(5)
    qs = [torch.randn(2, 32, generator=g), torch.randn(4, 32, generator=g)]
    ps = [torch.randn(8, 32, generator=g), torch.randn(4, 32, generator=g), torch.randn(16, 32, generator=g)]
    s_list = P.score_multi_vector(qs, ps, device="cpu").numpy()
    s_tens = P.score_multi_vector(torch.nn.utils.rnn.pad_sequence(qs, batch_first=True),
                                  torch.nn.utils.rnn.pad_sequence(ps, batch_first=True), device="cpu").numpy()
    save("score_fp32_d32.npz", q_lens=np.array([2, 4], np.int32), p_lens=np.array([8, 4, 16], np.int32),
         q=np.concatenate([q.numpy() for q in qs]), p=np.concatenate([p.numpy() for p in ps]),
         scores_list=s_list, scores_tensor=s_tens)

    # (6) losses: fp32 inputs, loss value + autograd gradients from the reference modules
    g = torch.Generator().manual_seed(99)
    B, C, Lq, Ld, dim = 6, 12, 9, 24, 128
    Q = F.normalize(torch.randn(B, Lq, dim, generator=g), dim=-1)
    Q[2, 6:] = 0  # padded query rows: lengths = (Q[:,:,0] != 0).sum(1)  (late_interaction_losses.py:296)
    D = F.normalize(torch.randn(C, Ld, dim, generator=g), dim=-1)
    D[5, 18:] = 0
    # make positives actually positive so top-2 / where logic sees both branches
    for b in range(B):
        for off in (0, 6):
            D[b + off, :Lq] = F.normalize(Q[b] + 0.4 * torch.randn(Lq, dim, generator=g), dim=-1) * (Q[b].abs().sum(-1, keepdim=True) > 0)
    D[1] = F.normalize(torch.randn(Ld, dim, generator=g), dim=-1)  # query 1's positive is NOT its best doc
    out = dict(Q=Q.numpy(), D=D.numpy())
    variants = {
        "default": dict(),
        "nonorm": dict(normalize_scores=False),
        "nonorm_T05": dict(normalize_scores=False, temperature=0.5),
        "filter": dict(normalize_scores=False, pos_aware_negative_filtering=True),
    }
    for cls_name in ("ColbertPairwiseCELoss", "ColbertLoss"):
        cls = getattr(L, cls_name)
        for vname, kw in variants.items():
            if cls_name == "ColbertLoss" and vname not in ("default", "nonorm"):
                continue
            for offset in (0, 6):
                q = Q.clone().requires_grad_(True)
                d = D.clone().requires_grad_(True)
                loss = cls(**kw)(q, d, offset=offset)
                loss.backward()
                key = f"{cls_name}_{vname}_off{offset}"
                out[key + "_loss"] = loss.detach().numpy()
                out[key + "_dQ"] = q.grad.numpy()
                out[key + "_dD"] = d.grad.numpy()
    # bf16 forward values of the same (for the dtype contract: bf16 in -> bf16 scalar out)
    for offset in (0, 6):
        out[f"ColbertPairwiseCELoss_nonorm_off{offset}_loss_bf16"] = (
            L.ColbertPairwiseCELoss(normalize_scores=False)(Q.bfloat16(), D.bfloat16(), offset=offset).float().numpy())
    save("loss_small.npz", **out)

    # (6b) explicit-negative variants on the same Q/D plus 2 negatives per query
    g = torch.Generator().manual_seed(123)
    N = F.normalize(torch.randn(B, 2, 20, dim, generator=g), dim=-1)
    N[0, 1, 15:] = 0
    outn = dict(Q=Q.numpy(), D=D.numpy(), N=N.numpy())
    for cls_name in ("ColbertNegativeCELoss", "ColbertPairwiseNegativeCELoss"):
        for vname, kw in {"default": dict(), "nonorm_w0": dict(normalize_scores=False, in_batch_term_weight=0.0),
                          "T1_w03": dict(temperature=1.0, in_batch_term_weight=0.3)}.items():
            for offset in (0, 6):
                q = Q.clone().requires_grad_(True); d = D.clone().requires_grad_(True); n = N.clone().requires_grad_(True)
                loss = getattr(L, cls_name)(**kw)(q, d, n, offset=offset)
                loss.backward()
                key = f"{cls_name}_{vname}_off{offset}"
                outn[key + "_loss"] = loss.detach().numpy()
                outn[key + "_dQ"] = q.grad.numpy()
                outn[key + "_dN"] = n.grad.numpy()
    save("loss_negatives.npz", **outn)

    # (7) helper known-answer tests restated from tests/loss/test_li_losses.py:45-73,137-147
    z = L.ColbertPairwiseCELoss(temperature=1.0, normalize_scores=False)(torch.zeros(2, 1, 3), torch.zeros(2, 1, 3))
    save("loss_kat.npz", pairwise_zero=z.numpy(), ln2=np.float32(np.log(2.0)))


def smooth_golden():
    """(8) use_smooth_max=True (late_interaction_losses.py:40-44, :88-90): loss + autograd gradients of the reference
    modules on the loss_small / loss_negatives inputs.  Separate file so the other fixtures stay byte-identical:
        python tests/golden/make_golden.py smooth"""
    zs = np.load(os.path.join(HERE, "loss_small.npz"))
    zn = np.load(os.path.join(HERE, "loss_negatives.npz"))
    Q, D, N = torch.from_numpy(zs["Q"]), torch.from_numpy(zs["D"]), torch.from_numpy(zn["N"])
    out = {}
    variants = {"tau01": dict(use_smooth_max=True), "tau002_nonorm_T1": dict(use_smooth_max=True, tau=0.02, normalize_scores=False,
                                                                          temperature=1.0)}
    for cls_name in ("ColbertPairwiseCELoss", "ColbertLoss"):
        for vname, kw in variants.items():
            for offset in (0, 6):
                q = Q.clone().requires_grad_(True); d = D.clone().requires_grad_(True)
                loss = getattr(L, cls_name)(**kw)(q, d, offset=offset)
                loss.backward()
                key = f"{cls_name}_{vname}_off{offset}"
                out[key + "_loss"] = loss.detach().numpy(); out[key + "_dQ"] = q.grad.numpy()
                if offset == 6:   # keep the fixture small: document gradients for one offset only
                    out[key + "_dD"] = d.grad.numpy()
    for cls_name in ("ColbertNegativeCELoss", "ColbertPairwiseNegativeCELoss"):
        for vname, kw in {"tau01": dict(use_smooth_max=True), "tau05_T1_w03": dict(use_smooth_max=True, tau=0.5, temperature=1.0,
                                                                                     in_batch_term_weight=0.3)}.items():
            for offset in (0, 6):
                q = Q.clone().requires_grad_(True); d = D.clone().requires_grad_(True); n = N.clone().requires_grad_(True)
                loss = getattr(L, cls_name)(**kw)(q, d, n, offset=offset)
                loss.backward()
                key = f"{cls_name}_{vname}_off{offset}"
                out[key + "_loss"] = loss.detach().numpy(); out[key + "_dQ"] = q.grad.numpy()
                if offset == 0:
                    out[key + "_dN"] = n.grad.numpy()
    save("loss_smooth.npz", **out)


SIGMOID_VARIANTS = {"default": dict(), "nonorm_T1": dict(normalize_scores=False, temperature=1.0),
                    "filter_T05": dict(pos_aware_negative_filtering=True, temperature=0.5),
                    "smooth_T1": dict(use_smooth_max=True, temperature=1.0)}


def sigmoid_golden():
    """(9) ColbertSigmoidLoss (late_interaction_losses.py:401-465) on the square in-batch case (C == B, offset 0), the only
    shape its flattened pos_mask construction (:456-462) addresses consistently.
        python tests/golden/make_golden.py sigmoid"""
    zs = np.load(os.path.join(HERE, "loss_small.npz"))
    Q, D = torch.from_numpy(zs["Q"]), torch.from_numpy(zs["D"])[:6].contiguous()
    out = {}
    for vname, kw in SIGMOID_VARIANTS.items():
        q = Q.clone().requires_grad_(True); d = D.clone().requires_grad_(True)
        loss = L.ColbertSigmoidLoss(**kw)(q, d)
        loss.backward()
        out[f"{vname}_loss"] = loss.detach().numpy(); out[f"{vname}_dQ"] = q.grad.numpy(); out[f"{vname}_dD"] = d.grad.numpy()
    save("loss_sigmoid.npz", **out)


def head_golden():
    """(10) the embedding head (modeling_colpali.py:67-72) as executed by the live reference's ColPali.forward on a
    random-init tiny PaliGemma config: the hidden states its custom_text_proj received, its weight / bias, the attention
    mask and the forward's return value, in fp32 and in bf16.
        python tests/golden/make_golden.py head"""
    from transformers import PaliGemmaConfig

    ColPali = refimport.load_colpali_class()
    cfg = PaliGemmaConfig(
        vision_config=dict(model_type="siglip_vision_model", hidden_size=32, intermediate_size=64, num_hidden_layers=1,
                           num_attention_heads=2, image_size=28, patch_size=14, projection_dim=128, vocab_size=300),
        text_config=dict(model_type="gemma", hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                         num_attention_heads=2, num_key_value_heads=1, head_dim=64, vocab_size=300),
        image_token_index=299, projection_dim=128, hidden_size=128, vocab_size=300)
    torch.manual_seed(0)
    model = ColPali(cfg).eval()
    g = torch.Generator().manual_seed(1)
    B, S = 5, 37
    ids = torch.randint(0, 290, (B, S), generator=g)
    mask = torch.ones(B, S, dtype=torch.long)
    mask[1, 30:] = 0          # right padding (ColPali style)
    mask[3, 11:] = 0
    out = {"input_ids": ids.numpy(), "attention_mask": mask.numpy()}
    for tag, dtype in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        m = model.to(dtype)
        cap = {}
        hook = m.custom_text_proj.register_forward_hook(lambda mod, inp, o: cap.update(h=inp[0].detach().clone()))
        with torch.no_grad():
            y = m(input_ids=ids, attention_mask=mask)
        hook.remove()
        conv = (lambda t: t.float().numpy()) if dtype == torch.float32 else (lambda t: bits(t.contiguous()))
        out[f"hidden_{tag}"] = conv(cap["h"])
        out[f"weight_{tag}"] = conv(m.custom_text_proj.weight.detach())
        out[f"bias_{tag}"] = conv(m.custom_text_proj.bias.detach())
        out[f"out_{tag}"] = conv(y)
        # the same forward inside the training graph (modeling_colpali.py:65-78 is part of it): autograd's gradients of the
        # head's weight / bias and of the hidden states it received, for the upstream gradient G (colpali_amd.embedding_head's
        # backward is pinned on these)
        gG = torch.Generator().manual_seed(2)
        G = torch.randn(B, S, 128, generator=gG).to(dtype)
        cap2 = {}

        def keep_grad(mod, inp):
            h = inp[0].detach().clone().requires_grad_(True)
            cap2["h"] = h
            return (h,)

        hook = m.custom_text_proj.register_forward_pre_hook(keep_grad)
        m.zero_grad(set_to_none=True)
        y2 = m(input_ids=ids, attention_mask=mask)
        hook.remove()
        assert torch.equal(y2.detach(), y)
        (y2 * G).sum().backward()
        out[f"gout_{tag}"] = conv(G)
        out[f"dweight_{tag}"] = conv(m.custom_text_proj.weight.grad)
        out[f"dbias_{tag}"] = conv(m.custom_text_proj.bias.grad)
        out[f"dhidden_{tag}"] = conv(cap2["h"].grad)
    save("head_colpali_tiny.npz", **out)


def sim_golden():
    """(11) score_single_vector (processing_utils.py:103-130) and get_similarity_maps_from_embeddings
    (interpretability/similarity_map_utils.py:9-55) of the live reference.
        python tests/golden/make_golden.py sim"""
    sm = refimport.load_similarity_map_utils()
    g = torch.Generator().manual_seed(31)
    out = {}
    # bi-encoder scores: fp32 (the reference's unit-test dtype, dim 32) and bf16 (BiPali width 1024)
    q32, p32 = torch.randn(4, 32, generator=g), torch.randn(8, 32, generator=g)
    out["sv_q_f32"], out["sv_p_f32"] = q32.numpy(), p32.numpy()
    out["sv_scores_f32"] = P.score_single_vector(list(q32), list(p32), device="cpu").numpy()
    qb, pb = unit_rows(37, 1024, g), unit_rows(101, 1024, g)
    out["sv_q_bf16"], out["sv_p_bf16"] = bits(qb), bits(pb)
    out["sv_scores_bf16"] = P.score_single_vector(qb, pb, device="cpu").numpy()                 # literal (bf16-rounded dots)
    out["sv_scores_bf16_truth"] = P.score_single_vector(qb.float(), pb.float(), device="cpu").numpy()
    # similarity maps: 2 images, 6 x 5 and 4 x 7 patch grids inside 40-token sequences, bf16 and fp32
    B, S, dim = 2, 40, 128
    img = torch.stack([unit_rows(S, dim, g) for _ in range(B)])
    qry = torch.stack([unit_rows(9, dim, g) for _ in range(B)])
    mask = torch.zeros(B, S, dtype=torch.bool)
    mask[0, 3:33] = True       # 30 = 6 x 5
    mask[1, 10:38] = True      # 28 = 4 x 7
    n_patches = [(6, 5), (4, 7)]
    out["map_img_bf16"], out["map_qry_bf16"], out["map_mask"] = bits(img), bits(qry), mask.numpy()
    out["map_n_patches"] = np.array(n_patches, np.int32)
    for tag, conv in (("bf16", lambda t: t), ("f32", lambda t: t.float())):
        maps = sm.get_similarity_maps_from_embeddings(conv(img), conv(qry), n_patches, mask)
        for i, m in enumerate(maps):
            out[f"map_{tag}_{i}"] = m.float().numpy()
    save("sim_matrix.npz", **out)


def pooling_golden():
    """(12) HierarchicalTokenPooler (compression/token_pooling/hierarchical_token_pooling.py:83-146) of the live reference:
    list inputs in fp32 and bf16, pool factors 2 / 3 / 4, one page with duplicated patches (exact ties).
        python tests/golden/make_golden.py pooling"""
    import warnings

    warnings.simplefilter("ignore")
    Pooler = refimport.load_token_pooler()
    g = torch.Generator().manual_seed(77)
    lens = [40, 97, 12, 2, 130]
    embs = [F.normalize(torch.randn(n, 128, generator=g), dim=-1) for n in lens]
    embs[1][10:30] = embs[1][5]                      # a blank region: identical patch embeddings
    # cluster-structured page: patches are noisy copies of a few prototypes (what real pages look like)
    proto = F.normalize(torch.randn(9, 128, generator=g), dim=-1)
    embs.append(F.normalize(proto[torch.randint(0, 9, (150,), generator=g)] + 0.15 * torch.randn(150, 128, generator=g), dim=-1))
    out = {"lens": np.array([e.shape[0] for e in embs], np.int32), "emb_f32": np.concatenate([e.numpy() for e in embs])}
    for tag, conv in (("f32", lambda t: t), ("bf16", lambda t: t.to(torch.bfloat16))):
        for pf in (2, 3, 4):
            res = Pooler().pool_embeddings([conv(e) for e in embs], pool_factor=pf, return_dict=True)
            for i, (pe, mp) in enumerate(zip(res.pooled_embeddings, res.cluster_id_to_indices)):
                out[f"{tag}_pf{pf}_{i}_pooled"] = pe.float().numpy()
                labels = np.full(embs[i].shape[0], -1, np.int32)
                for c, idx in mp.items():
                    labels[idx[0].numpy()] = c
                out[f"{tag}_pf{pf}_{i}_labels"] = labels
    save("token_pooling.npz", **out)


def planted_inputs(seed, n_q, Lq, n_d, Ld_lo, Ld_hi, n_planted, dim=128):
    """Planted-retrieval inputs: unit-row random queries and documents; for every query `n_planted` documents
    carry a noisy copy of each of its tokens at a random row.  The noise vector of planted copy j has norm 0.3 .. 0.75
    (cosine 0.96 .. 0.80), so the planted documents out-score every random one (score ~ 9) by a wide margin and each other
    by ~0.5: the top-`n_planted` ranking is decided far above any summation-order noise.  Regenerated from the seed by the
    tests (tests/helpers.py:planted_inputs restates this function); the sha256 pins the torch RNG stream."""
    g = torch.Generator().manual_seed(seed)
    qs = [unit_rows(Lq, dim, g) for _ in range(n_q)]
    lens = [Ld_lo] * n_d if Ld_lo == Ld_hi else torch.randint(Ld_lo, Ld_hi + 1, (n_d,), generator=g).tolist()
    ps = [unit_rows(n, dim, g).float() for n in lens]
    slots = torch.randperm(n_d, generator=g)[: n_q * n_planted].view(n_q, n_planted)
    for qi in range(n_q):
        for j in range(n_planted):
            d = int(slots[qi, j])
            sigma = 0.3 + 0.05 * j
            noisy = F.normalize(qs[qi].float() + sigma * torch.randn(Lq, dim, generator=g) / dim**0.5, dim=-1)
            rows = torch.randperm(lens[d], generator=g)[:Lq]
            ps[d][rows] = noisy
    ps = [p.to(torch.bfloat16) for p in ps]
    return qs, ps, slots


def topk_golden():
    """(13) end-to-end ranking (north star: "bit-exact top-k doc indices"): the live reference's fp32 scorer
    (processing_utils.py:132-187 on .float() copies, device="cpu") followed by torch.topk, as scripts/compute_hardnegs.py:92-94
    (k = 100) and processing_utils.py:189-219 (k = 10) rank.  Two planted corpora, inputs regenerated from the seed:
        python tests/golden/make_golden.py topk"""
    out = {}
    for tag, (seed, n_q, Lq, n_d, lo, hi, n_pl) in {"dense": (4242, 8, 32, 2000, 256, 256, 10),
                                                    "ragged": (4343, 8, 32, 1000, 267, 779, 10)}.items():
        qs, ps, slots = planted_inputs(seed, n_q, Lq, n_d, lo, hi, n_pl)
        h = hashlib.sha256()
        for t in qs + ps:
            h.update(bits(t).tobytes())
        scores = P.score_multi_vector([q.float() for q in qs], [p.float() for p in ps], batch_size=10**9, device="cpu")
        blocked = P.score_multi_vector([q.float() for q in qs], [p.float() for p in ps], device="cpu")   # default blocking
        out[f"{tag}_params"] = np.array([seed, n_q, Lq, n_d, lo, hi, n_pl], np.int64)
        out[f"{tag}_sha256"] = np.frombuffer(h.digest(), dtype=np.uint8)
        out[f"{tag}_scores"] = scores.numpy()
        out[f"{tag}_scores_bs128"] = blocked.numpy()
        out[f"{tag}_planted"] = slots.numpy()
        for k in (10, 100):
            out[f"{tag}_top{k}"] = scores.topk(k, dim=1).indices.numpy()
            out[f"{tag}_top{k}_bs128"] = blocked.topk(k, dim=1).indices.numpy()
        top10 = scores.topk(10, dim=1)
        gaps = (top10.values[:, :-1] - top10.values[:, 1:]).min()
        print(tag, "min gap inside the planted top-10:", float(gaps), "10th vs 11th:",
              float((scores.topk(11, dim=1).values[:, 9] - scores.topk(11, dim=1).values[:, 10]).min()))
        assert sorted(top10.indices[0].tolist()) == sorted(slots[0].tolist())
    save("topk_planted.npz", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "topk":
        topk_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "pooling":
        pooling_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "sim":
        sim_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "head":
        head_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "smooth":
        smooth_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "sigmoid":
        sigmoid_golden()
    else:
        main()
        smooth_golden()
        sigmoid_golden()
        head_golden()
        sim_golden()
        pooling_golden()
        topk_golden()
