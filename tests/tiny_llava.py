"""A LLaVA-NeXT checkpoint small enough to build in a test: the stock `transformers` classes the reference loads
(`LlavaNextProcessor`, `LlavaNextForConditionalGeneration`, vla_system/llava_processor.py:18-31) with a 2-layer CLIP tower, a
2-layer Mistral decoder, a 21-word vocabulary and RANDOM weights, saved to a local directory (no hub access, no download).
It exercises the real path -- processor -> pixel values + image tokens -> generate (greedy, 10 new tokens) -> decode -> parse --
on whatever device / dtype the scorer is given; what the model "says" is noise, which is exactly the reference's
unparseable-answer case unless the noise happens to be a number."""
import numpy as np
import torch

WORDS = ["<pad>", "<s>", "</s>", "<unk>", "<image>"] + list("0123456789.") + ["assistant", "user", "system", "Task", "Rate"]


def build_tiny_llava(path, seed=0):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import (CLIPVisionConfig, LlavaNextConfig, LlavaNextForConditionalGeneration, LlavaNextProcessor,
                              MistralConfig, PreTrainedTokenizerFast)
    try:   # torchvision is not installed in this image: the PIL implementation of the image processor
        from transformers.models.llava_next.image_processing_pil_llava_next import LlavaNextImageProcessorPil as ImageProc
    except ImportError:   # older transformers
        from transformers import LlavaNextImageProcessor as ImageProc
    vocab = {w: i for i, w in enumerate(WORDS)}
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tk = PreTrainedTokenizerFast(tokenizer_object=tok, pad_token="<pad>", bos_token="<s>", eos_token="</s>", unk_token="<unk>",
                                 additional_special_tokens=["<image>"])
    pins = [[28, 28], [56, 28], [28, 56]]
    ip = ImageProc(size={"shortest_edge": 28}, crop_size={"height": 28, "width": 28}, image_grid_pinpoints=pins)
    proc = LlavaNextProcessor(image_processor=ip, tokenizer=tk, patch_size=14, vision_feature_select_strategy="default",
                              image_token="<image>", num_additional_image_tokens=1)
    vc = CLIPVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, image_size=28,
                          patch_size=14, projection_dim=32)
    tc = MistralConfig(vocab_size=len(WORDS), hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                       num_key_value_heads=2, max_position_embeddings=4096, pad_token_id=0, bos_token_id=1, eos_token_id=2)
    cfg = LlavaNextConfig(vision_config=vc, text_config=tc, image_token_index=vocab["<image>"], image_grid_pinpoints=pins,
                          vision_feature_layer=-1)
    torch.manual_seed(seed)
    model = LlavaNextForConditionalGeneration(cfg).eval()
    model.save_pretrained(path)
    proc.save_pretrained(path)
    return path


def sample_image(h=48, w=64, seed=0):
    return np.random.default_rng(seed).integers(0, 255, (h, w, 3), dtype=np.uint8)
