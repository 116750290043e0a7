"""Group the steady-state kernel list of tools/prof_step.sh (gpurun_out/prof_step/s_kernel_stats.csv) into families."""
import collections
import csv
import sys

NS = 6.0
f = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_step/s_kernel_stats.csv'
g = collections.Counter(); c = collections.Counter()
for r in csv.DictReader(open(f)):
    n = r['Name']; t = float(r['TotalDurationNs']) / 1e6 / NS; k = int(r['Calls']) / NS
    if n.startswith('Cijk'):
        fam = 'hipBLASLt GEMM, fp32 out (wgrad)' if '_BSS_' in n or '_SB_' in n else 'hipBLASLt GEMM, bf16 out'
    elif 'xfm::deep' in n and ('ILi1EEE' in n or ', 1>' in n or 'fwd1' in n): fam = 'shallow-fusion scan (single-route kernels, d_state 16)'
    elif 'xfm::deep' in n: fam = 'SS2D deep-fusion scan, backward (d_state 16)'
    elif 'ss2dc_fwd_kernel<7, 16' in n: fam = 'SS2D deep-fusion scan, forward (d_state 16)'
    elif 'xfm::ss2dc' in n or 'xfm::chan_' in n or 'xfm::chan1' in n: fam = 'SS2D channel-lane scan (+post)'
    elif 'lean' in n or 'ss2d_l3' in n or 'ss2d_w_' in n or 'dt_proj' in n or 'route_' in n: fam = 'SS2D wide-map scan + dt_proj + route split/merge'
    elif 'rowscan' in n or 'swap' in n or 'selective_scan' in n or 'xfm::scan_' in n: fam = 'shallow-fusion scan (single-route kernels, d_state 16)'
    elif 'at::native' in n: fam = 'framework reduce' if 'reduce_kernel' in n else 'framework elementwise / copy / fill'
    elif 'igemm' in n or 'miopen' in n.lower() or 'ck::' in n or '_ZN2ck' in n or 'batched_transpose' in n or 'SubTensor' in n or 'naive_conv' in n: fam = 'MIOpen convolution (+ its casts)'
    elif 'partial_sums' in n: fam = 'deferred column sums (LayerNorm / bias gradients, folded per step)'
    elif 'rowln' in n or 'settle_' in n: fam = 'row LayerNorm (+residual)'
    elif 'ln2d' in n: fam = 'LayerNorm2d'
    elif 'conv_im2col' in n or 'conv_col2im' in n: fam = 'own 3x3 stride-2 convolutions: neighbourhood rows (their GEMMs: the own MFMA families)'
    elif 'wgrad_kernel' in n or 'wgrad_tt' in n: fam = 'own MFMA weight-gradient GEMM'
    elif 'tokens_gemm3_kernel<0' in n or 'tokens_gemm2_kernel<' in n and ', 0, ' in n: fam = 'own MFMA GEMM'
    elif 'tokens_gemm2' in n or 'tokens_gemm3' in n: fam = 'own MFMA GEMM with GELU epilogue (Mlp fc1 / fc2 data gradient)'
    elif 'tokens_gemm' in n or 'planes_gemm' in n or 'proj_gemm' in n or 'proj_tiled' in n: fam = 'own MFMA GEMM'
    elif 'transpose_short' in n: fam = 'tokens <-> planes transposes (7x7)'
    elif 'bn_tokens' in n: fam = 'BatchNorm (two views, token-major)'
    elif 'batch_norm' in n or 'CatArray' in n or 'elementwise_kernel' in n or 'rocclr' in n or 'layer_norm' in n or 'GammaBeta' in n or 'distribution_' in n or 'multi_tensor_apply' in n or 'softmax' in n or 'nll_loss' in n:
        fam = 'framework elementwise / copy / fill'
    elif 'tokens_kernel' in n or 'colsum' in n: fam = 'bias+GELU / column sums'
    elif 'dwconv' in n: fam = 'depthwise conv + SiLU'
    elif 'adam' in n: fam = 'fused Adam'
    else: fam = 'other'
    g[fam] += t; c[fam] += k
tot = sum(g.values())
print(f"{'family':52s} launches/step  ms/step   share")
for fam, t in g.most_common(): print(f"{fam:52s} {c[fam]:10.1f} {t:9.3f} {100 * t / tot:6.1f}%")
print(f"{'total':52s} {sum(c.values()):10.1f} {tot:9.3f}")
