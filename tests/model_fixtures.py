"""Tiny random-init instances of the REFERENCE's Col* model classes (oracle/refimport.py: the live checkout in the build container,
the fetched git-ignored copy under tests/_reference_pkg/ on the GPU box).  Test infrastructure; no weights exist offline."""
import torch

from oracle import refimport


def tiny_colpali(mask_non_image_embeddings=False, seed=0):
    from transformers import PaliGemmaConfig

    ColPali = refimport.load_model_class("models/paligemma/colpali/modeling_colpali", "ColPali")
    cfg = PaliGemmaConfig(
        vision_config=dict(model_type="siglip_vision_model", hidden_size=32, intermediate_size=64, num_hidden_layers=1,
                           num_attention_heads=2, image_size=28, patch_size=14, projection_dim=128, vocab_size=300),
        text_config=dict(model_type="gemma", hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                         num_attention_heads=2, num_key_value_heads=1, head_dim=64, vocab_size=300),
        image_token_index=299, projection_dim=128, hidden_size=128, vocab_size=300)
    torch.manual_seed(seed)
    return ColPali(cfg, mask_non_image_embeddings=mask_non_image_embeddings).eval(), ColPali


def tiny_colqwen2(seed=0):
    from transformers import Qwen2VLConfig

    ColQwen2 = refimport.load_model_class("models/qwen2/colqwen2/modeling_colqwen2", "ColQwen2")
    cfg = Qwen2VLConfig(
        text_config=dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                         vocab_size=300, rope_scaling={"type": "mrope", "mrope_section": [8, 12, 12]}, max_position_embeddings=512,
                         bos_token_id=1, eos_token_id=2),
        vision_config=dict(depth=1, embed_dim=32, hidden_size=128, num_heads=2, mlp_ratio=2, patch_size=14, spatial_merge_size=2,
                           temporal_patch_size=2, in_channels=3),
        image_token_id=299, video_token_id=298, vision_start_token_id=297, vision_end_token_id=296, vocab_size=300,
        bos_token_id=1, eos_token_id=2)
    torch.manual_seed(seed)
    return ColQwen2(cfg).eval(), ColQwen2


def colpali_page_batch(B=4, n_text=9, seed=1, device="cpu"):
    """ColPali-style inputs WITH an image: 4 image tokens (28 x 28 image, patch 14) + text, right padding."""
    g = torch.Generator().manual_seed(seed)
    S = 4 + n_text
    ids = torch.randint(0, 290, (B, S), generator=g)
    ids[:, :4] = 299
    mask = torch.ones(B, S, dtype=torch.long)
    mask[1, S - 3:] = 0
    pix = torch.randn(B, 3, 28, 28, generator=g)
    return dict(input_ids=ids.to(device), attention_mask=mask.to(device), pixel_values=pix.to(device))


def text_batch(B=5, S=37, seed=2, left_pad=False, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, 290, (B, S), generator=g)
    mask = torch.ones(B, S, dtype=torch.long)
    if left_pad:
        mask[1, :S // 5] = 0
        mask[B - 1, :S // 2] = 0
    else:
        mask[1, S - S // 5:] = 0
        mask[B - 1, S // 3:] = 0
    return dict(input_ids=ids.to(device), attention_mask=mask.to(device))


def colqwen2_page_batch(model, B=3, h=4, w=4, n_text=6, seed=3, device="cpu"):
    """ColQwen2-style inputs WITH an image, as ColQwen2Processor hands them over: `pixel_values` padded per image beyond its h x w
    patches (the forward un-pads with image_grid_thw, modeling_colqwen2.py:50-56), h*w/4 image tokens between the vision markers,
    left padding, mm_token_type_ids (1 = image token: transformers' M-RoPE needs it)."""
    cfg = model.config
    g = torch.Generator().manual_seed(seed)
    n_img = h * w // 4
    S = 2 + n_img + n_text + 2
    ids = torch.randint(0, 290, (B, S), generator=g)
    mask = torch.ones(B, S, dtype=torch.long)
    mask[1, :2] = 0                                  # left padding in front of the vision block
    ids[:, 2] = cfg.vision_start_token_id
    ids[:, 3:3 + n_img] = cfg.image_token_id
    ids[:, 3 + n_img] = cfg.vision_end_token_id
    pix = torch.randn(B, h * w + 5, 3 * 2 * 14 * 14, generator=g)
    return dict(input_ids=ids.to(device), attention_mask=mask.to(device), pixel_values=pix.to(device),
                image_grid_thw=torch.tensor([[1, h, w]] * B, device=device), mm_token_type_ids=(ids == cfg.image_token_id).int().to(device))
