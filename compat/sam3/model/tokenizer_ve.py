"""``sam3.model.tokenizer_ve`` facade (reference: sam3/sam3/model/tokenizer_ve.py:128-253): the CLIP BPE tokenizer of
the text path under the reference's class name, constructor arguments and return type (a LongTensor [n, ctx])."""
import torch

from efficientsam3_amd.tokenizer import ClipBpeTokenizer


class SimpleTokenizer(ClipBpeTokenizer):
    def __init__(self, bpe_path, additional_special_tokens=None, context_length=77, clean="lower"):
        if additional_special_tokens:
            raise NotImplementedError("additional_special_tokens (no caller on the image path passes them)")
        if clean != "lower":
            raise NotImplementedError(f"clean={clean!r}: the reference's builders only use 'lower'")
        super().__init__(bpe_path, context_length=context_length)
        self.sot_token_id, self.eot_token_id = self.sot_id, self.eot_id
        self.vocab_size = len(self.token_id)

    def __call__(self, texts, context_length=None) -> torch.LongTensor:
        return torch.from_numpy(super().__call__(texts, context_length=context_length)).long()
