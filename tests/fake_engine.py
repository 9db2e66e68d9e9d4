"""TEST DOUBLE of efficientsam3_amd.engine.HipEngine, backed by the CPU oracle (oracle/ref_model.py).

Purpose: let the host side of the boundary -- the import facade (compat/sam3), the builder, Sam3Image, Sam3Processor,
checkpoint ingestion -- be exercised by the REFERENCE'S OWN CALLERS (eval/eval_coco.py) in the build container, which
has the reference but no GPU.  It is never used by the product: the real HipEngine raises without a HIP device.
Only the image path (preprocess / encode / decode / postprocess / clamp_) is provided."""
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from oracle import ref_model


class OracleEngine:
    def __init__(self, backbone_type="efficientvit", model_name="b1", dtype="f32", device=None, interactive=True,
                 fuse_linear_chains=True):
        self.backbone_type, self.model_name, self.interactive = backbone_type, model_name, interactive
        self.device = torch.device("cpu")
        self.torch_dtype = torch.float32
        self.dtype_name = "f32"
        self.finalized = False
        self.sd = None

    def load_state_dict(self, sd):
        self.sd = dict(sd)

    def finalize(self):
        self.finalized = True

    def set_text_causal(self, causal):
        pass

    def preprocess_u8(self, img_hwc_u8: torch.Tensor) -> torch.Tensor:
        return torch.stack([ref_model.normalise_image_u8(t.permute(2, 0, 1)) for t in img_hwc_u8])

    def preprocess_resize_u8(self, img_hwc_u8: torch.Tensor, out_chw: torch.Tensor) -> torch.Tensor:
        out_chw.copy_(ref_model.processor_transform(img_hwc_u8.permute(2, 0, 1).contiguous()))
        return out_chw

    def encode(self, img_nchw, want_sam3=True, want_sam2=True, want_trunk=False, want_stages=False, out=None):
        with torch.inference_mode():
            fo = ref_model.forward_image(self.sd, img_nchw.float(), self.model_name)
        res = {}
        if want_sam3:
            res["sam3_fpn"] = [t.permute(0, 2, 3, 1).contiguous() for t in fo["backbone_fpn"]]
        if want_sam2:
            res["sam2_fpn"] = [t.permute(0, 2, 3, 1).contiguous() for t in fo["sam2_backbone_out"]["backbone_fpn"]]
        return res

    def decode(self, sam2_fpn: Sequence[torch.Tensor], prompt_image, coords, labels, multimask_output, want_obj=False,
               mask_input: Optional[torch.Tensor] = None, out=None):
        """the part of ref_model.predict_inst between the prompt arithmetic and the post-processing, per prompt set"""
        R = ref_model
        sd = self.sd
        fpn = [t.permute(0, 3, 1, 2) for t in sam2_fpn]
        lows, ious = [], []
        with torch.inference_mode():
            for j in range(prompt_image.shape[0]):
                i = int(prompt_image[j])
                image_embed = fpn[2][i:i + 1] + sd[R.SAM + "no_mem_embed"].view(1, -1, 1, 1)
                if coords is not None:
                    sparse = R.embed_points(sd, coords[j:j + 1].float(), labels[j:j + 1].float(), pad=True)
                else:
                    sparse = torch.empty((1, 0, 256))
                if mask_input is not None:
                    dense = R.embed_masks(sd, mask_input[j:j + 1, None].float())
                else:
                    dense = sd[R.SAM + "sam_prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(1, -1, R.EMB, R.EMB)
                low, iou, _ = R.mask_decoder(sd, image_embed, sparse, dense, fpn[0][i:i + 1], fpn[1][i:i + 1],
                                             multimask_output, False, None)
                lows.append(low[0])
                ious.append(iou[0])
        return torch.stack(lows).float(), torch.stack(ious).float()

    def postprocess(self, low_res, orig_hw: Tuple[int, int], return_logits, max_hole_area=256.0, mask_threshold=0.0, out=None):
        lead = low_res.shape[:-2]
        with torch.inference_mode():
            m = ref_model.postprocess_masks(low_res.reshape(-1, 1, *low_res.shape[-2:]).float(), orig_hw)
        m = m.reshape(*lead, int(orig_hw[0]), int(orig_hw[1]))
        return m if return_logits else (m > mask_threshold).to(torch.uint8)

    def clamp_(self, x, lo, hi):
        return x.clamp_(lo, hi)
