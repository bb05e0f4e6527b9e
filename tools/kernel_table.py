#!/usr/bin/env python3
"""The kernel-to-launch table at the top of DESIGN.md, from a warm per-kernel statistics file (``tools/quick_stats.sh`` /
``tools/collect_profiles.sh`` -> ``profiles/<tag>_kernel_stats.csv``): every launch of one training step at 8 x 64 grouped by what it
does, with the engine method that enqueues it, the C-ABI entry point and its share of the step.

    python tools/kernel_table.py profiles/r06_final_kernel_stats.csv        (prints markdown)
"""
import csv
import re
import sys

# (row label, where it is enqueued, C-ABI entry point(s), bound, [kernel-name regexes])   - first match wins, in this order
ROWS = [
    ("conv3 forward over the listed windows (f16, ReLU + pool epilogue, rows gathered by the window list)", "engine_fwd.conv3_shared",
     "sgc_conv3_relu_pool_windows_wm", "MFMA", [r"gemm_nt_pp_kernel<0, 3, 0, 1"]),
    ("conv3 data gradient, real pairs' windows, patch form on the sparse matrix cores", "engine_bwd._conv3_backward_shared",
     "sgc_windows_dgrad_patches_sparse", "MFMA (sparse) / operand arrival", [r"gemm_nt_sp_kernel"]),
    ("conv3 weight gradient, real pairs' windows, sparse matrix cores (second operand gathered from the f16 maps)", "engine_bwd._conv3_backward_shared",
     "sgc_windows_wgrad_gather_sparse / _patch_sparse", "MFMA (sparse) / operand arrival", [r"gemm_tn_sp_kernel<[12]>"]),
    ("fc1 over the window-major rows: forward products", "engine_fwd.fc1_shared", "sgc_fc1_windows_gemm[_x16]", "MFMA, K = 1024",
     [r"gemm_nt_pp_kernel<0, [78], 0, 0", r"gemm_nt_pp_kernel<0, 7"]),
    ("fc1 weight gradient (grouped TN) + conv2 / conv3-object weight gradients", "engine_bwd._fc1_backward_rows / train_backward",
     "sgc_fc1_windows_wgrad, sgc_conv2_wgrad, sgc_conv3_wgrad[_sparse], sgc_windows_wgrad_patch", "MFMA", [r"gemm_tn_pp_kernel", r"gemm_tn_sp_kernel<0>", r"gemm_tn_kernel"]),
    ("fc1 data gradient (grouped NT), conv3 data gradient of the dense tail / objects, fc2, conv2, conv1 GEMMs", "engine_bwd / engine_fwd",
     "sgc_fc1_windows_dgrad, sgc_windows_dgrad_patches, sgc_conv3_dgrad[_pooled], sgc_fc2_*, sgc_conv2_*, sgc_conv1_tanh", "MFMA",
     [r"gemm_nt_pp_kernel", r"gemm_nt_kernel", r"conv16_halo"]),
    ("pair expansion z = maxpool(relu(U_i + V_j)) next to the pair-specific windows (+ pseudo-pairs)", "engine_fwd.expand", "sgc_pair_expand_dense_windows, sgc_pair_expand_train",
     "vector instructions, then HBM", [r"pair_expand"]),
    ("pair contraction dU_i = sum_j g_ij (both roles)", "engine_bwd.train_backward", "sgc_pair_contract_windows", "vector instructions, then HBM", [r"pair_contract"]),
    ("fc1 assembly / prefix sums / row sums / X rows", "engine_fwd.fc1_shared, engine_bwd._fc1_backward_rows", "sgc_fc1_assemble*, sgc_fc1_integral, sgc_fc1_gsum, sgc_fc1_xrows",
     "HBM / L2", [r"fc1_assemble", r"fc1_integral", r"fc1_gsum", r"fc1_xrows", r"fc1_own_rect", r"segment_sum_rows"]),
    ("window backward glue: patch sums, packers of the sparse operands, patch copy of the dense tail, un-pool of the tail", "engine_bwd._conv3_backward_shared",
     "sgc_windows_patch_sum2, sgc_windows_dgrad_sparse_pack, (pack inside the sparse wgrad call), sgc_windows_im2patch_f16[_from], sgc_windows_unpool_from", "HBM",
     [r"windows_patch_sum", r"nt_sp_pack", r"windows_sparse_pack", r"windows_im2patch", r"windows_unpool", r"unpool"]),
    ("linear pairs (windows combined, not convolved) forward / backward, background maps", "engine_fwd.conv3_shared, engine_bwd", "sgc_windows_linear_*, sgc_shared_objects_bg_grad",
     "HBM / latency", [r"windows_linear", r"shared_bg_grad"]),
    ("optimizer: fused fc1 update (un-permute + SGD-momentum + f16 copy), multi-tensor SGD", "optim.FusedSGD.step", "sgc_sgd_fc1_fused, sgc_sgd_momentum_multi, sgc_sgd_momentum_step",
     "HBM", [r"sgd_"]),
    ("weight copies (16-bit layouts of the updated weights), casts, transposes", "engine_weights", "sgc_permute_cast, sgc_transpose_cast, sgc_segment_cast, sgc_windows_dgrad_sparse_weights, sgc_convert_f16_bf16",
     "HBM", [r"permute_cast", r"transpose_cast", r"segment_cast", r"nt_sp_weights", r"convert_f16_bf16", r"pack_nhwc"]),
    ("split-K slab sums, column sums (bias gradients)", "engine_bwd._slab_sum / _colsum", "sgc_slab_sum, sgc_colsum", "HBM / latency", [r"slab_sum", r"colsum"]),
    ("head: logits + Bayesian product + candidates; loss gradient; head weight gradient; label vectors", "engine_fwd.head, engine_bwd.train_backward",
     "sgc_bayes_head, sgc_head_loss_bwd, sgc_head_wgrad, sgc_label_*", "latency", [r"bayes_head", r"head_loss_bwd", r"head_wgrad", r"label_", r"connectivity_stats", r"loss_coefficients"]),
    ("masks: per-object masked maps forward / backward, tanh backward, conv2 region fills", "engine_fwd.object_halves, engine_bwd", "sgc_object_masked_maps[_bwd], sgc_tanh_bwd, sgc_conv2_*regions*",
     "HBM / latency", [r"mask_objects", r"tanh_bwd", r"conv2_fill", r"conv2_regions", r"conv2_bwd_regions", r"fill_zero"]),
    ("scene tables + shared-window plan (pair tables, window lists, placement)", "pairs.flatten_scene, engine_plan", "sgc_scene_tables, sgc_shared_windows_*, sgc_bucket_place_seg, sgc_scan_rows",
     "latency", [r"scene_", r"shared_", r"bucket_", r"scan_rows", r"window_rows", r"pseudo_pair"]),
    ("torch-native elementwise / cat / reduce launches still issued from Python, memcpy / memset", "various", "-", "latency", [r"at::native", r"rocclr", r"cub::", r"rocprim"]),
]


def main(path):
    rows = []
    for r in csv.reader(l for l in open(path) if not l.startswith("#")):
        if len(r) >= 6 and r[0].isdigit():
            rows.append((int(r[0]), float(r[1]), float(r[4]), r[5]))
    steps = 4                                               # tools/quick_stats.sh: statistics over the last 4 steps
    acc = [[0, 0.0, []] for _ in ROWS]
    other = [0, 0.0, []]
    for calls, total, per_step, name in rows:
        for k, row in enumerate(ROWS):
            if any(re.search(p, name) for p in row[4]):
                acc[k][0] += calls / steps; acc[k][1] += per_step; acc[k][2].append(name)
                break
        else:
            other[0] += calls / steps; other[1] += per_step; other[2].append(name)
    tot = sum(a[1] for a in acc) + other[1]
    print("| what | enqueued by | C-ABI entry | bound | launches / step | ms / step | share |")
    print("|---|---|---|---|---|---|---|")
    for row, a in zip(ROWS, acc):
        if a[0]:
            print("| %s | `%s` | `%s` | %s | %.0f | %.2f | %.0f %% |" % (row[0], row[1], row[2], row[3], a[0], a[1], 100 * a[1] / tot))
    if other[0]:
        print("| other: %s | | | | %.0f | %.2f | %.0f %% |" % (", ".join(sorted({n.split("(")[0][:40] for n in other[2]}))[:200], other[0], other[1], 100 * other[1] / tot))
    print("| **all launches of one warm step (single stream)** | | | | **%.0f** | **%.2f** | |" % (sum(a[0] for a in acc) + other[0], tot))


if __name__ == "__main__":
    main(sys.argv[1])
