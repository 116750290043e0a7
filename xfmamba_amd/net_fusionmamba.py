"""``TwoViewXFMambaTop`` -- the model the upstream README calls ``dualfusionmambav13``.

Same constructor, ``forward(x_a, x_b)`` and ``state_dict`` keys as ``net_fusionmamba.py:141-210``
of XZheng0427/XFMamba; the blocks come from ``xfmamba_amd.fusion_vmamba`` and run the HIP
kernels.  The ablation models of the reference file are out of scope.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from .fusion_vmamba import Backbone_VSSM, CSSFVSSLayer_v5, ShallowFusionBlock_v4

__all__ = ["TwoViewXFMambaTop", "ModelWrapper"]

# XFM_STACKED_FUSION=0: the two fusion blocks on NCHW maps, one call per view pair as the reference writes them (read once)
STACKED_FUSION = os.environ.get("XFM_STACKED_FUSION", "1") == "1"

_TRUNKS = {   # net_fusionmamba.py:151-159
    "small": dict(depths=[2, 2, 15, 2], dims=96, drop_path_rate=0.3, ssm_ratio=2.0),
    "base": dict(depths=[2, 2, 15, 2], dims=128, drop_path_rate=0.6, ssm_ratio=2.0),
    "tiny": dict(depths=[2, 2, 8, 2], dims=96, drop_path_rate=0.2, ssm_ratio=1.0),
}


class ModelWrapper(nn.Module):
    """Feeds a channel-concatenated pair to a two-input model (net_fusionmamba.py:10-26)."""

    def __init__(self, original_model, output_index=0):
        super().__init__()
        self.model = original_model
        self.output_index = output_index

    def forward(self, input_tensor):
        assert input_tensor.size(1) % 2 == 0, "The channel dimension must be even to split into two inputs."
        c = input_tensor.size(1) // 2
        out = self.model(input_tensor[:, :c], input_tensor[:, c:])
        return out[self.output_index] if isinstance(out, (tuple, list)) else out


class TwoViewXFMambaTop(nn.Module):
    def __init__(self, in_channels, outputs, attention_downsampling=4, hidden_dim=768, depth=1, attn_drop_rate=0.,
                 d_state=16, drop_path_rate=0.1, pretrained=None, type='small'):
        super().__init__()
        assert in_channels == 1, 'in_channels expected to be 1'
        if type not in _TRUNKS:
            raise ValueError(f"type must be one of {sorted(_TRUNKS)}")
        self.mamba_feature_extrac = Backbone_VSSM(pretrained=pretrained, **_TRUNKS[type])
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, depth)]
        self.shallow_mamba_fusion = ShallowFusionBlock_v4(hidden_dim=hidden_dim, attn_drop_rate=attn_drop_rate,
                                                          d_state=d_state)
        # depth is hard-wired to 1 upstream whatever `depth` says (net_fusionmamba.py:170-177)
        self.fusemamba = CSSFVSSLayer_v5(hidden_dim=hidden_dim, depth=1, drop_path=dpr, attn_drop_rate=attn_drop_rate,
                                         d_state=d_state, attention_downsampling=attention_downsampling)
        self.final_conv = nn.Conv2d(hidden_dim, hidden_dim, kernel_size=1)
        self.classifier = nn.Sequential(OrderedDict(avgpool=nn.AdaptiveAvgPool2d(1), flatten=nn.Flatten(1),
                                                    head=nn.Linear(hidden_dim, outputs)))
        # The trunk has no cross-sample op (LayerNorm only), so running both views as one batch of 2B
        # gives the same features as the reference's two sequential calls and halves the launches.
        self.merge_views = True

    def _stacked_ok(self, x_a) -> bool:
        tr = self.mamba_feature_extrac
        return (STACKED_FUSION and self.merge_views and x_a.is_cuda and (len(tr.layers) - 1) in tr.out_indices
                and tr.tokens_path_ok(x_a))

    def _head(self, z, tokens: bool):
        """``classifier(final_conv(z))``.  A 1x1 convolution commutes with the spatial mean (mean_l (W z_l + b) =
        W mean_l z_l + b), so with the reference's avgpool -> flatten -> Linear classifier the convolution runs on the
        pooled (B, C) rows: 49 times less work than on the map, same fp32 arithmetic up to summation order."""
        fc, cls = self.final_conv, self.classifier
        pooled = (isinstance(getattr(cls, "avgpool", None), nn.AdaptiveAvgPool2d) and cls.avgpool.output_size in (1, (1, 1))
                  and isinstance(getattr(cls, "flatten", None), nn.Flatten) and len(cls) == 3
                  and fc.kernel_size == (1, 1) and fc.stride == (1, 1) and fc.padding == (0, 0) and fc.groups == 1
                  and fc.padding_mode == "zeros")
        if not pooled:
            return cls(fc(z.permute(0, 3, 1, 2) if tokens else z))
        zp = z.mean((1, 2) if tokens else (2, 3))
        return cls.head(F.linear(zp, fc.weight.view(fc.out_channels, fc.in_channels), fc.bias))

    def forward(self, x_a, x_b):
        if self._stacked_ok(x_a):
            # the trunk's token-major stream goes on through both fusion blocks: [view 1 | view 2] (2B, H, W, C)
            zt = self.mamba_feature_extrac(torch.cat([x_a, x_b], dim=0).expand(-1, 3, -1, -1), only_last=True,
                                           tokens_out=True)[-1]
            if self.shallow_mamba_fusion.stacked_ok(zt) and self.fusemamba.stacked_ok(zt):
                # (A = -exp(A_logs) of the two blocks stays two tiny kernels each: batched through _NegExpAll the multi-tensor
                #  launches of two tensors measured 21 + 10 + 17 us against 4 x 2.6 + 4 x 4.8)
                z = self.fusemamba.forward_stacked(self.shallow_mamba_fusion.forward_stacked(zt))
                with torch.autocast("cuda", enabled=False):
                    return self._head(z.float(), tokens=True)
            z = zt.permute(0, 3, 1, 2).contiguous()
            z_a, z_b = z.chunk(2, dim=0)
        elif self.merge_views:
            # (concatenate the 1-channel views first: the 3-channel broadcast stays a stride-0 view for the trunk)
            z = self.mamba_feature_extrac(torch.cat([x_a, x_b], dim=0).expand(-1, 3, -1, -1), only_last=True)[-1]
            z_a, z_b = z.chunk(2, dim=0)
        else:
            x_a = x_a.expand(-1, 3, -1, -1)
            x_b = x_b.expand(-1, 3, -1, -1)
            z_a = self.mamba_feature_extrac(x_a)[3]
            z_b = self.mamba_feature_extrac(x_b)[3]
        z_a, z_b = self.shallow_mamba_fusion(z_a, z_b)
        z = self.fusemamba(z_a, z_b)
        # the 768->768->outputs head is negligible work: keep it out of autocast so the logits are not quantised
        # to the 8-bit bf16 mantissa (no effect when autocast is off, i.e. on the fp32 reference path)
        with torch.autocast("cuda", enabled=False):
            return self._head(z.float(), tokens=False)
