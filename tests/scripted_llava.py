"""A scripted stand-in for the (LlavaNextProcessor, LlavaNextForConditionalGeneration) pair: no weights exist on this
filesystem, so the live path of the LLaVA scorer -- processor(prompt, image) -> generate -> decode -> parse
(vla_system/llava_processor.py:79-101) -- is driven with texts a chat model could return.  Used by
tests/golden/make_golden_r3.py to record what the REFERENCE class does with them, and by the tests to drive this
repo's LLaVAScorer with the same script."""
import torch


class ScriptedInputs(dict):
    def to(self, device):
        return self


class ScriptedProcessor:
    """processor(prompt, image, return_tensors=...) -> inputs; decode(ids, skip_special_tokens=True) -> text."""
    def __init__(self, script):
        self.script, self.prompts, self.n = script, [], 0

    def __call__(self, *args, text=None, images=None, return_tensors="pt"):
        # the reference calls processor(prompt, image, ...) positionally (the transformers 4.3x signature); this repo's scorer
        # passes text= / images= by keyword, which every transformers version accepts
        prompt = args[0] if args else text
        self.prompts.append(prompt)
        i = self.n
        self.n += 1
        return ScriptedInputs(input_ids=torch.tensor([[i]]))

    def decode(self, ids, skip_special_tokens=True):
        return self.script[int(ids[0])]


class ScriptedModel:
    def __init__(self, script):
        self.script, self.kwargs = script, []

    def generate(self, **kw):
        self.kwargs.append({k: v for k, v in kw.items() if k != "input_ids"})
        i = int(kw["input_ids"][0, 0])
        if self.script[i] == "!raise":
            raise RuntimeError("scripted generate failure")
        return kw["input_ids"]
