/* C-ABI of libsgc_relhead.so: the MI355X (gfx950) kernels of the pairwise relation-prediction path.
 *
 * The reference (bowen-upenn/scene_graph_commonsense) is pure PyTorch and has no FFI; every entry point
 * below replaces a *stock PyTorch op sequence* of its hot path and cites it (paths relative to the reference
 * repo).  Conventions:
 *   - all pointers are DEVICE pointers unless stated; the library owns no memory and never synchronises;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); kernels are enqueued on it;
 *   - return 0 = ok, 1 = bad argument (shape/alignment contract violated), 2 = launch error;
 *   - "f16"/"bf16" tensors are raw 16-bit storage; accumulation is always f32;
 *   - layouts are channels-last.  Pixel rows of conv outputs are "window-major": row m of an SxS map is
 *     m = 4*(py*(S/2)+px) + (dy*2+dx), i.e. the four pixels of a 2x2 pooling window are adjacent.
 * Sizes are specialised to the reference defaults hidden_dim=128 (D), feature_size=32 (F): C = 2D+1 = 257 input
 * channels zero-padded to XC=384, conv2 512 ch, conv3 1024 ch, fc1 65536->4096, fc2 ->512 (other sizes: the sgc_generic_* trunk
 * at the end of this file).
 * The Python binding is scene_graph_commonsense_amd/_lib.py (ctypes); INTEGRATION.md shows the call sites.
 */
#ifndef SGC_RELHEAD_H
#define SGC_RELHEAD_H
#ifdef __cplusplus
extern "C" {
#endif

/* ----------------------------------------------------------------------------------------------- forward */

/* cat(image_feature, image_depth) NCHW f32 -> f16 [n_img*HW][XC]   (train_test.py:194-195 `torch.cat((feat*mask, depth*mask), dim=1)`;
 * the mask is applied later, after conv1).  f1 may be NULL with C1 = 0 (already concatenated 257-channel input). */
int sgc_pack_image_nhwc(const float* f0, int C0, const float* f1, int C1, void* x_out, int n_img, int HW, int XC, void* stream);

/* a_img [n_rows][128] f16 = tanh(x [n_rows][XC] * w1r[128][XC]^T + b1)   (model.py:139-140, conv1_1 / conv1_2 + tanh; once per image and role) */
int sgc_conv1_tanh(const void* x, const void* w1r, const float* b1, void* a_img, int n_rows, int XC, void* stream);

/* a_pad [n_obj][F+2][F+2][D] f16: inside bbox (x0,x1,y0,y1, slice semantics) the image map, outside tanh(b1), zero border
 * (train_test.py:164-169 mask build + :194-195,202-203 masked gather; tanh(conv1(0)) = tanh(bias) outside the box). */
int sgc_object_masked_maps(const void* a_img, const int* obj_img, const int* bbox, const void* cst, void* a_pad, int n_obj, int F, int D, void* stream);

/* U or V [n_obj*1024][512] f16 = conv3x3(a_pad, w2r[512][2][9][64]) (+bias)   (model.py:141-143: conv2_1 over the channel concat
 * is the sum of a subject half U_i and an object half V_j; bias goes with V). */
int sgc_conv2_object(const void* a_pad, const void* w2r, const float* bias, void* out, int n_obj, void* stream);

/* z_pad [n_pairs][18][18][512] = maxpool2(relu(U[sub_idx] + V[obj_idx]))   (model.py:143-144 ReLU + maxpool; the pair expansion).
 * out_elem: 0 = f16, 1 = bf16. */
int sgc_pair_expand(const void* U, const void* V, const int* sub_idx, const int* obj_idx, void* z_pad, int n_pairs, int out_elem, void* stream);

/* Dense form of the same expansion for ALL ordered pairs of every image (the fused path): one workgroup per (image, window,
 * tile of 16 objects) stages the V quads in LDS, every wavefront keeps one subject's U quad in registers and writes complete 1 KiB
 * channel rows.  pid [n_obj][pid_ld] maps (subject object, object index inside the image) -> pair index (-1 = skip);
 * img_ptr [n_img+1]; any output may be NULL. */
int sgc_pair_expand_dense(const void* U, const void* V, const int* img_ptr, const int* pid, int pid_ld, int n_img, int max_n,
                          void* z_pad_f16, void* z_pad_bf16, unsigned char* amz, void* stream);
/* The same, writing only the pixels pixel_rect[pair] names (sgc_shared_windows_count): with conv3 over shared windows a pair's
 * expansion is read only next to its pair-specific windows; NULL = every pixel. */
int sgc_pair_expand_dense_windows(const void* U, const void* V, const int* img_ptr, const int* pid, int pid_ld, int n_img, int max_n,
                                  void* z_pad_f16, void* z_pad_bf16, unsigned char* amz, const int* pixel_rect, void* stream);

/* y [n_pairs*64][1024] f16 (+ argmax u8, may be NULL; + y_bf16, the same values rounded to bf16 for the fc1 weight gradient, may be
 * NULL) = maxpool2(relu(conv3x3(z_pad, w3r[1024][8][9][64]) + b3))   (model.py:145-146) */
int sgc_conv3_relu_pool(const void* z_pad, const void* w3r, const float* b3, void* y, unsigned char* argmax, void* y_bf16, int n_pairs,
                        void* stream);

/* conv3 over SHARED windows (csrc/kernels_shared.hip).  Outside its box an object's masked map is the constant tanh(b1)
 * (train_test.py:194-195), so the part of conv3_1 -> ReLU -> max-pool (model.py:145-147) that depends on BOTH objects of a pair is
 * confined to the intersection X_ij = R_i n R_j of two rectangles of conv3's 8x8 pooling-window grid; every other window of the pair
 * equals the same window of the pseudo-pair (i, empty-box object) or (empty-box object, j), computed once per object.
 * bbox [n_obj][4] int (x0,x1,y0,y1, slice semantics); sub_idx / obj_idx [n_pairs] index it.
 *   sgc_shared_windows_count   count[p] = |X_p|; pixel_rect[p] (may be NULL) = the 16-grid pixels within one pixel of X_p, packed
 *                              Y0 | Y1<<5 | X0<<10 | X1<<15 (0 = none): where the pair's z, routing codes and dz are needed
 *   sgc_shared_windows_fill    gather[e] = pair*64 + window for all X windows, given the INCLUSIVE prefix sum of count
 *   sgc_conv3_relu_pool_windows  sgc_conv3_relu_pool for the listed windows only, results written to rows gather[e] of y / argmax /
 *                              y_bf16 (max_entries bounds the launch, *gather_n on the device is the list length)
 *   sgc_shared_windows_assemble  the remaining rows: copies of y_obj [2*n_obj*64][1024] (rows of pseudo-pair (i, bg) = i,
 *                              of (bg, j) = n_obj + j), likewise argmax_obj and y_obj_bf16 (each may be NULL with its output) */
int sgc_shared_windows_count(const int* bbox, const int* sub_idx, const int* obj_idx, int n_pairs, int* count, int* pixel_rect,
                             void* stream);
int sgc_shared_windows_fill(const int* bbox, const int* sub_idx, const int* obj_idx, int n_pairs, const int* count_incl, int* gather,
                            void* stream);
int sgc_conv3_relu_pool_windows(const void* z_pad, const void* w3r, const float* b3, const int* gather, const int* gather_n,
                                int max_entries, void* y, unsigned char* argmax, void* y_bf16, void* stream);
int sgc_shared_windows_assemble(const int* bbox, const int* sub_idx, const int* obj_idx, int n_pairs, int n_obj, const void* y_obj,
                                const unsigned char* argmax_obj, const void* y_obj_bf16, void* y, unsigned char* argmax, void* y_bf16,
                                void* stream);

/* conv2_1 halves on the objects' own regions (model.py:143 on the masked maps of train_test.py:194-195): outside its box an object's
 * masked map is the constant tanh(b1), so U_o / V_o equal the background's at every pixel whose 3x3 neighbourhood misses the box.
 *   sgc_conv2_regions_count / _fill   the 2x2-pixel windows of the 32-grid that can differ: count[o], gather[e] = o*256 + wy*16 + wx
 *   sgc_conv2_object_regions          sgc_conv2_object for the listed windows, rows written to their window-major place in uv
 *   sgc_conv2_fill_background         every other row of objects 0 .. n_obj-1 = the row of the background object n_obj + obj_img[o]
 *                                     (uv [n_obj + n_img][1024][512] f16; the background objects' maps are computed whole beforehand) */
int sgc_conv2_regions_count(const int* bbox, int n_obj, int* count, void* stream);
int sgc_conv2_regions_fill(const int* bbox, int n_obj, const int* count_incl, int* gather, void* stream);
int sgc_conv2_object_regions(const void* a_pad, const void* w2r, const float* bias, const int* gather, const int* gather_n,
                             int max_entries, void* uv, void* stream);
int sgc_conv2_fill_background(const int* bbox, const int* obj_img, int n_obj, void* uv, void* stream);

/* Linear pairs (csrc/kernels_shared.hip, "sixth identity"): when the 16-grid regions where two objects' conv2 halves differ from the
 * background's are disjoint, z_ij = z_(i,bg) + z_(bg,j) - z_(bg,bg) pixel by pixel, and conv3_1 (model.py:145, linear up to its ReLU)
 * gives pre_ij = pre_(i,bg) + pre_(bg,j) - pre_(bg,bg) on the pair's X windows: no convolution of its own, combined from the
 * pre-activations of the per-object window entries; backward = the un-pooled gradient added to those entries, subtracted from the
 * image's background map.
 *   sgc_shared_windows_count3          count_all[p] = |X_p|, count_conv[p] / count_linear[p] = |X_p| for the pairs of that class (else
 *                                      0), pixel_rect_conv[p] = sgc_shared_windows_count's rectangle, empty for a linear pair
 *   sgc_shared_windows_fill_class      sgc_shared_windows_fill for one class (cls 1 = not linear, 2 = linear) and its prefix counts
 *   sgc_conv3_windows_raw              raw [4*max_entries][1024] f32 = conv3x3(z_pad, w3r) of the listed windows, no bias / ReLU / pool
 *   sgc_windows_linear_forward         y / y_bf16 (row dest_all[entry in the list of ALL X windows]) and argmax (row pair*64 + window) of
 *                                      the linear pairs' windows from raw = [per-object entries in list order | 64 windows of every
 *                                      image's background map]; count_incl_all over [pairs | 2 n_obj pseudo-pairs]
 *   sgc_windows_linear_backward_objects  dy3x rows of the n_object_entries per-object entries (list positions first_object_entry ..) +=
 *                                      un-pooled gradient of the linear pairs' windows that used them
 *   sgc_windows_linear_backward_bg     dy3_bg_pad [n_img][18][18][1024] -= the same per (image, window); order / segments = the linear
 *                                      list sorted by image*64 + window and its 64 n_img + 1 range starts; dest_linear from the
 *                                      forward; bias_part [64 n_img][1024] */
int sgc_shared_windows_count3(const int* bbox, const int* sub_idx, const int* obj_idx, int n_pairs, int* count_all, int* count_conv,
                              int* count_linear, int* pixel_rect_conv, void* stream);
int sgc_shared_windows_fill_class(const int* bbox, const int* sub_idx, const int* obj_idx, int n_pairs, const int* count_incl, int* gather,
                                  int cls, void* stream);
int sgc_conv3_windows_raw(const void* z_pad, const void* w3r, const int* gather, const int* gather_n, int max_entries, float* raw,
                          void* stream);
/* sgc_conv3_relu_pool_windows_wm that also stores the accumulators (no bias / ReLU / pooling) of the entries e >= raw_first - the
 * per-object entries at the tail of the list - to raw[(e - raw_first)*4 + pixel][1024] f32: no second launch for them */
int sgc_conv3_relu_pool_windows_wm_raw(const void* z_pad, const void* w3r, const float* b3, const int* gather, const int* gather_n,
                                       const int* dest, int max_entries, void* ywm, unsigned char* argmax, void* ywm_bf16, float* raw,
                                       int raw_first, void* stream);
int sgc_windows_linear_forward(const int* bbox, const int* sub_idx, const int* obj_idx, const int* obj_img, int n_obj, int n_real_pairs,
                               const int* gather_linear, const int* n_linear, int max_linear, const int* count_incl_all,
                               const int* dest_all, const float* raw, long n_object_entries, const float* b3, void* ywm, void* ywm_bf16,
                               unsigned char* argmax, int* dest_linear, void* stream);
int sgc_windows_linear_backward_objects(const int* bbox, const int* sub_idx, const int* obj_idx, const int* sub_ptr, const int* sub_list,
                                        const int* obj_ptr, const int* obj_list, int n_obj, int n_real_pairs, const int* gather_conv,
                                        int first_object_entry, int n_object_entries, const int* count_incl_all, const int* dest_all,
                                        const void* dywm, const unsigned char* argmax, void* dy3x, void* stream);
int sgc_windows_linear_backward_bg(const int* gather_linear, const int* dest_linear, const int* order, const int* segments, int n_img,
                                   const void* dywm, const unsigned char* argmax, void* dy3_bg_pad, float* bias_part, void* stream);

/* Backward of the shared-window conv3 (autodiff of the graph above; same citations).  All gradients bf16, sums f32.
 *   sgc_shared_windows_assemble_bwd  dy_obj [2*n_obj*64][1024] = per-object sums of the rows of dy [n_pairs*64][1024] that were copies
 *                                    (I windows -> subject's row, J windows -> object's row); sub_ptr/sub_list, obj_ptr/obj_list =
 *                                    CSR lists of the pairs of every object in each role (sgc_scene_tables)
 *   sgc_windows_unpool               dy3x [4*entries_pad][1024]: ReLU + max-pool backward of the listed (X) windows, row 4e+q = pixel q
 *                                    of window gather[e]; rows of entries >= *gather_n are zero; bias_part [*n_parts <= 1024][1024];
 *                                    the gradient row of entry e is dy[dest[e]] (window-major, shared fc1) or dy[gather[e]] (dest NULL)
 *   sgc_windows_im2col               zcol [4*entries_pad][9][512] = the 3x3 neighbourhoods of those pixels in z_pad_bf16
 *   sgc_windows_wgrad                conv3 weight-gradient slabs [splits][1024][9*512] of the X windows = dy3x^T * zcol
 *   sgc_windows_dgrad_cols           col [rows][9][512] = dy3x * w3col^T, w3col [(tap, c_in)][c_out] bf16
 *   sgc_windows_col2im               dz [pair*256 + pixel][512] = sum over taps of col, for the pixels within one pixel of a pair's X
 *                                    windows (the other rows of a real pair's dz are never written nor read)
 *   sgc_pair_contract_windows        sgc_pair_contract over those rows + the pseudo-pairs (pair index n_real_pairs + o for (o, bg),
 *                                    n_real_pairs + n_obj + o for (bg, o)); dU_pad has n_obj + n_img objects, object n_obj + b = the
 *                                    background of image b (objects img_ptr[b] .. img_ptr[b+1]); pixel_rect [n_real_pairs + 2*n_obj
 *                                    (+ n_img)]: sgc_shared_windows_count for the real pairs, the whole map (16 << 5 | 16 << 15) or
 *                                    sgc_shared_objects_count for the pseudo-pairs */
int sgc_shared_windows_assemble_bwd(const int* bbox, const int* sub_idx, const int* obj_idx, const int* sub_ptr, const int* sub_list,
                                    const int* obj_ptr, const int* obj_list, int n_obj, const void* dy, void* dy_obj, void* stream);
int sgc_windows_unpool(const void* dy, const unsigned char* argmax, const int* gather, const int* gather_n, const int* dest,
                       int entries_pad, void* dy3x, float* bias_part, int* n_parts, void* stream);
int sgc_windows_im2col(const void* z_pad_bf16, const int* gather, const int* gather_n, int entries_pad, void* zcol, void* stream);
int sgc_windows_wgrad(const void* dy3x, const void* zcol, float* slabs, int rows, int splits, int* n_slabs, void* stream);
/* ... without the im2col buffer: the rows of the second operand are gathered from z_pad_bf16 through the window list; gather must hold
 * valid windows of fully written maps for all rows/4 entries (the rows of dy3x behind the list are zero) */
int sgc_windows_wgrad_gather(const void* dy3x, const void* z_pad_bf16, const int* gather, float* slabs, int rows, int splits, int* n_slabs,
                             void* stream);
/* PATCH form of the weight gradient over the listed windows (the step's default): zpatch [entries_pad][16][512] bf16 = the 4 x 4 input
 * patch of every listed window (zero rows behind the list), read by the weight-gradient product at (own pixel + tap) - 16 instead of 36
 * rows per window (reference: the autograd of model.py:144-146's conv3 w.r.t. its weight). */
int sgc_windows_im2patch(const void* z_pad_bf16, const int* gather, const int* gather_n, int entries_pad, void* zpatch, void* stream);
int sgc_windows_im2patch_f16(const void* z_pad_f16, const int* gather, const int* gather_n, int entries_pad, void* zpatch, void* stream);   /* source: the forward's f16 maps */
int sgc_windows_wgrad_patch(const void* dy3x, const void* zpatch, float* slabs, int rows, int splits, int* n_slabs, void* stream);
/* The same product for the first n_entries (a multiple of 16) listed windows on the SPARSE matrix cores (v_smfmac_f32_32x32x32_bf16:
 * the gradient before ReLU + 2x2 max-pool has one non-zero per window and channel, windows are 4 consecutive K indices - the 2:4
 * pattern; model.py:145-147 backward).  Valid for the real pairs' windows only (the per-object entries behind them hold sums of
 * several windows' gradients: sgc_windows_wgrad_patch on that tail).  dywm: POOLED gradient rows, row dest[e]; argmax at gather[e];
 * pack_ac (n_entries/16 * 64 KiB) / pack_ic (n_entries/16 * 8 KiB): scratch for the packed operand. */
/* splits: > 0 that many K ranges (split-K slabs); 0: chosen for whole rounds of blocks on the 256 CUs; -1: K ranges PER XCD (32 ranges, every
 * XCD runs all tiles of one channel half for every fourth range - the operands leave the fabric once / twice instead of four / two times;
 * needs n_entries >= 32768, else as 0).  *n_slabs returns the count written. */
int sgc_windows_wgrad_patch_sparse(const void* dywm, const unsigned char* argmax, const int* gather, const int* dest, int n_entries,
                                   const void* zpatch, void* pack_ac, void* pack_ic, float* slabs, int splits, int* n_slabs, void* stream);
/* ... with the second operand gathered from the forward's f16 maps z_pad_f16 [pairs][18][18][512] through the window list inside the GEMM
 * block (no patch copy of these windows; f16 -> bf16 in registers: the same bits as sgc_windows_im2patch_f16 + the call above), and the
 * patch copy of a TAIL of the list (entries e0 .. e0 + entries - 1, zpatch row 0 = entry e0) for the dense block behind it. */
int sgc_windows_wgrad_gather_sparse(const void* dywm, const unsigned char* argmax, const int* gather, const int* dest, int n_entries,
                                    const void* z_pad_f16, void* pack_ac, void* pack_ic, float* slabs, int splits, int* n_slabs, void* stream);
int sgc_windows_im2patch_f16_from(const void* z_pad_f16, const int* gather, const int* gather_n, int e0, int entries, void* zpatch, void* stream);
int sgc_windows_dgrad_cols(const void* dy3x, const void* w3col, void* col, int rows, void* stream);
int sgc_windows_col2im(const void* col, const int* bbox, const int* sub_idx, const int* obj_idx, const int* count_incl, int n_pairs,
                       void* dz, void* stream);
/* The same data gradient in PATCH form (the step's default): patch [entries][slots][512] bf16, slots = sgc_windows_patch_slots() = 20:
 * the gradient of the 4 x 4 input patch (pixels 2wy-1 .. 2wy+2, 2wx-1 .. 2wx+2) of every listed window, summed over the taps inside the
 * GEMM (K = 1024 x the 1 / 2 / 4 (own pixel, tap) combinations that reach a patch pixel; the four centre pixels as two rows of two
 * combinations each, so that K <= 2048); w3patch as engine.prep_bwd_weights lays it out.  sgc_windows_patch_sum[_objects]: dz of a
 * pixel = sum of the rows of the <= 2 x 2 windows of the pair's rectangle that cover it (same output as sgc_windows_col2im[_objects];
 * reference: the autograd of model.py:144-146's conv3). */
/* Data gradient of conv3 over the listed windows 0 .. n_sparse-1 (a multiple of 256; the real pairs' windows: one non-zero per window and
 * channel in the un-pooled gradient) on the SPARSE matrix cores - the patch form of sgc_windows_dgrad_patches (same products, the structural
 * zeros not issued; output patch [n_sparse][16][512] bf16: one row per patch pixel, see sgc_windows_patch_sum2); reference arithmetic: the backward of
 * model.py:145-147.  dywm: pooled gradient rows (row dest[e], or gather[e] when dest is NULL); argmax: routing bytes at gather[e];
 * w3sp [20][512][2048] bf16 from sgc_windows_dgrad_sparse_weights(conv3_1.weight f32 [1024][512][3][3]); sgc_windows_dgrad_sparse_pack
 * fills pack_a (4 * n_sparse * 2 KiB: the pooled rows masked to the four own-pixel sets) and pack_i (4 * n_sparse * 256 B: index words) and,
 * when bias_part is given, the conv3 bias partial sums of these windows ([*n_parts][1024], reduce with sgc_slab_sum).  The entries behind n_sparse (per-object entries: dense sums) take sgc_windows_unpool_from +
 * sgc_windows_dgrad_patches into the rows behind. */
int sgc_windows_dgrad_sparse_weights(const float* conv3_weight, void* w3sp, void* stream);
/* sgc_windows_patch_sum / _objects for a patch buffer whose first n16 entries are in the sparse form's layout - 16 rows of 512 per entry, one
 * per patch pixel (the centre pixels' two halves are summed in its accumulators) - followed by the dense form's 20 rows per entry for the
 * entries behind them (at patch + n16 * 16 * 512). */
int sgc_windows_patch_sum2(const void* patch, int n16, const int* bbox, const int* sub_idx, const int* obj_idx, const int* count_incl, int n_pairs,
                           void* dz, void* stream);
int sgc_windows_patch_sum_objects2(const void* patch, int n16, const int* bbox, int n_obj, int n_real, const int* count_incl, void* dz, void* stream);
int sgc_windows_dgrad_sparse_pack(const void* dywm, const unsigned char* argmax, const int* gather, const int* dest, int n_sparse,
                                  void* pack_a, void* pack_i, float* bias_part, int* n_parts, void* stream);
int sgc_windows_dgrad_patches_sparse(const void* pack_a, const void* pack_i, int n_sparse, const void* w3sp, void* patch, void* stream);
int sgc_windows_unpool_from(const void* dy, const unsigned char* argmax, const int* gather, const int* gather_n, const int* dest, int entry0,
                            int entries_pad, void* dy3x, float* bias_part, int* n_parts, void* stream);
int sgc_windows_patch_slots(void);
int sgc_windows_dgrad_patches(const void* dy3x, const void* w3patch, void* patch, int entries, void* stream);
int sgc_windows_patch_sum(const void* patch, const int* bbox, const int* sub_idx, const int* obj_idx, const int* count_incl, int n_pairs,
                          void* dz, void* stream);
int sgc_windows_patch_sum_objects(const void* patch, const int* bbox, int n_obj, int n_real_pairs, const int* count_incl, void* dz, void* stream);
int sgc_pair_contract_windows(const void* dz, const unsigned char* amz, const int* ptr, const int* list, const int* pixel_rect,
                              const int* img_ptr, int role, int n_real_pairs, int n_obj, int n_img, int bg_maps, void* dU_pad, void* stream);
/* Second level of the same sharing: a per-object map equals the all-background map outside R_o, so a pseudo-pair is computed only
 * on the windows of R_o - appended to the window list as entries of the "pair" n_real_pairs + ps (ps = role*n_obj + o) - and takes the
 * rest from the background map of its image (pair n_real_pairs + 2*n_obj + image).  pixel_rect of sgc_pair_contract_windows then has
 * n_real_pairs + 2*n_obj + n_img entries (pseudo-pairs: around R_o; background maps: everything) and bg_maps = 1.
 *   sgc_shared_objects_count      count[ps] = |R_o|, pixel_rect[ps] (arrays offset to the pseudo-pairs)
 *   sgc_shared_objects_fill       their entries of the window list (count_incl = inclusive prefix over real pairs + pseudo-pairs)
 *   sgc_shared_objects_fill_rows  window-major rows (and pair-major routing rows) of the pseudo-pairs outside R_o = copies of the
 *                                 background maps y_bg [n_img*64][1024]
 *   sgc_shared_objects_bg_grad    dy_bg [n_img*64][1024] = the sum of the gradient rows of those copies
 *   sgc_windows_col2im_objects    sgc_windows_col2im for the pseudo-pairs' entries */
int sgc_shared_objects_count(const int* bbox, int n_obj, int* count, int* pixel_rect, void* stream);
int sgc_shared_objects_fill(const int* bbox, int n_obj, int n_real_pairs, const int* count_incl, int* gather, void* stream);
int sgc_shared_objects_fill_rows(const int* bbox, const int* obj_img, int n_obj, const int* goff, const void* y_bg, const void* y_bg_bf16,
                                 const unsigned char* argmax_bg, void* ywm, void* ywm_bf16, unsigned char* argmax_ps, void* stream);
int sgc_shared_objects_bg_grad(const int* bbox, const int* img_ptr, int n_obj, int n_img, const int* goff, const void* dywm, void* dy_bg,
                               void* stream);
int sgc_windows_col2im_objects(const void* col, const int* bbox, int n_obj, int n_real_pairs, const int* count_incl, void* dz, void* stream);

/* fc1 over shared windows (csrc/kernels_shared.hip; model.py:148-149).  fc1 is a sum over conv3's 64 pooling windows and the I / J
 * windows of a pair are copies of per-object rows, so every row fc1 multiplies lives in one WINDOW-MAJOR row space: group w =
 * [2*n_obj pseudo-pair rows][X entries of window w in pair order][zero rows up to a multiple of 256]; goff [65] = group offsets.
 *   sgc_conv3_relu_pool_wm          sgc_conv3_relu_pool of the pseudo-pairs, y / y_bf16 rows written at goff[w] + pseudo-pair
 *   sgc_conv3_relu_pool_windows_wm  ... of the X windows, y / y_bf16 of entry e written at dest[e] (argmax at gather[e] as before)
 *   sgc_fc1_products_pitch          row pitch of owm in floats (4096 + padding: a power-of-two pitch slows the product's stores)
 *   sgc_fc1_windows_gemm            owm [rows][pitch] f32 (columns 0..4095) = ywm [rows][1024] * (columns g*1024.. of w1p)^T, g = tile_group[row/256]
 *   sgc_fc1_integral                S [n_pseudo][9][9][4096] = 2-D inclusive prefix sums (zero border) of the pseudo rows of owm
 *   sgc_fc1_own_rect_sums           own[j] = S'_j[R_j]: the rectangle term of the assembly that depends on one object only
 *   sgc_fc1_assemble                h1[p] = dropout(relu(b + S_i[all] - S_i[R_j] + S'_j[R_j] - S'_j[X_p] + sum_{e in X_p} owm[dest[e]])) */
int sgc_conv3_relu_pool_wm(const void* z_pad, const void* w3r, const float* b3, const int* goff, void* ywm, unsigned char* argmax,
                           void* ywm_bf16, int n_pairs, void* stream);
int sgc_conv3_relu_pool_windows_wm(const void* z_pad, const void* w3r, const float* b3, const int* gather, const int* gather_n,
                                   const int* dest, int max_entries, void* ywm, unsigned char* argmax, void* ywm_bf16, void* stream);
int sgc_fc1_products_pitch(void);
int sgc_fc1_windows_gemm(const void* ywm, const void* w1p, const int* tile_group, float* owm, int rows, void* stream);
int sgc_fc1_integral(const float* owm, const int* goff, int n_pseudo, float* S, void* stream);
int sgc_fc1_own_rect_sums(const float* S, const int* bbox, int n_obj, float* own, void* stream);   /* own [n_obj][4096] = S'_j[R_j] */
int sgc_fc1_assemble(const float* S, const float* owm, const int* bbox, const int* sub_idx, const int* obj_idx, const int* count_incl,
                     const int* dest, int n_obj, const float* bias, int drop_enable, unsigned drop_seed, void* h1, int n_pairs,
                     const float* own_rect_sums /* may be NULL: the four corner reads per pair */, void* stream);
/* The same rows, workgroup b assembling pair pair_order[b] (a permutation of the pairs, e.g. the contraction's list sorted by subject: the
 * subject's prefix table then stays L2-resident over its ~N-1 consecutive pairs; NULL = pair order).  Same bits. */
int sgc_fc1_assemble_ordered(const float* S, const float* owm, const int* bbox, const int* sub_idx, const int* obj_idx, const int* count_incl,
                             const int* dest, int n_obj, const float* bias, int drop_enable, unsigned drop_seed, void* h1, int n_pairs,
                             const float* own_rect_sums, const int* pair_order, void* stream);
/* fc1 over the window-major rows with the pair-specific (X) rows as f16: sgc_fc1_windows_gemm_x16 writes the rows whose index inside their
 * window group is >= n_pseudo (goff: the groups' first rows) to oxh [rows][4096] f16 and only the per-object rows in front of them to
 * owm (f32: their 2-D prefix sums follow); sgc_fc1_assemble_x16 reads the pairs' X products from oxh.  Reference: model.py:148-149. */
int sgc_fc1_windows_gemm_x16(const void* ywm, const void* w1p, const int* tile_group, const int* goff, int n_pseudo, float* owm, void* oxh,
                             int rows, void* stream);
int sgc_fc1_assemble_x16(const float* S, const void* oxh, const int* bbox, const int* sub_idx, const int* obj_idx, const int* count_incl,
                         const int* dest, int n_obj, const float* bias, int drop_enable, unsigned drop_seed, void* h1, int n_pairs,
                         const float* own_rect_sums, const int* pair_order, void* stream);

/* Backward of the shared fc1 (window-major rows; all bf16, f32 sums):
 *   sgc_fc1_gsum           pseudo rows of gwm [rows][4096]: row goff[w] + role*n_obj + o = sum of dh1 over the pairs of object o in that
 *                          role whose window w was a copy of o's row
 *   sgc_fc1_xrows          X rows: gwm[dest[e]] = dh1[pair of entry e]; zeroes the padding rows [gend[g], goff[g+1]) of gwm and ywm_bf16
 *   sgc_fc1_windows_dgrad  dywm [rows][1024] = gwm * (rows g*1024.. of w1pT [65536][4096])^T (per-object rows: gradient of the
 *                          pseudo-pairs' conv3 output; X rows: of the entries')
 *   sgc_fc1_windows_wgrad  dw [4096][65536] f32, column block g = gwm[group g]^T * ywm_bf16[group g] */
int sgc_fc1_gsum(const void* dh1, const int* bbox, const int* sub_idx, const int* obj_idx, const int* sub_ptr, const int* sub_list,
                 const int* obj_ptr, const int* obj_list, const int* goff, int n_obj, void* gwm, void* stream);
int sgc_fc1_xrows(const void* dh1, const int* gather, const int* dest, int n_entries, const int* goff, const int* gend, void* gwm,
                  void* ywm_bf16, void* stream);
int sgc_fc1_windows_dgrad(const void* gwm, const void* w1pT, const int* tile_group, void* dywm, int rows, void* stream);
int sgc_fc1_windows_wgrad(const void* gwm, const void* ywm_bf16, const int* goff, float* dw, int rows, void* stream);

/* h1 [n_pairs][4096] f16 = dropout(relu(y[n_pairs][K] * w1p[4096][K]^T + b))   (model.py:148-149; columns of w1p in (window, channel) order) */
int sgc_fc1_relu(const void* y, const void* w1p, const float* b, void* h1, int n_pairs, int K, int drop_enable, unsigned drop_seed, void* stream);

/* p [n_pairs][512] f32 = dropout(relu(h1 * w2m[512][4096]^T + b + Lsub[sub_idx] + Lobj[obj_idx]))
 * (model.py:152-168 one-hot/multi-hot concat + :175 fc2: the label columns of fc2 become per-object row vectors). */
int sgc_fc2_labels_relu(const void* h1, const void* w2m, const float* b, const float* lsub, const float* lobj, const int* sub_idx,
                        const int* obj_idx, float* p_out, int n_pairs, int drop_enable, unsigned drop_seed, void* stream);

/* Hierarchical Bayesian head + candidate reduction   (model.py:176-184 = BayesianHead model.py:24-34; flat: model.py:99-101;
 * candidates evaluator.py:160-174: per super-category max log-prob / first argmax, -inf where iou_mask == 0).
 * Wt [512][64] f32 (column r = output row r: R relation rows, then 3 super rows and the connectivity row (hier) or the
 * connectivity row (flat)), bias [64].  rel [n_pairs][R], sup [n_pairs][3] (hier), conn [n_pairs],
 * cand_conf/cand_pred [n_pairs][3] (hier) or [n_pairs] (flat).  iou_mask may be NULL. */
int sgc_bayes_head(const float* p, const float* Wt, const float* bias, int n_pairs, int ng, int np, int ns, int hier, float T1, float T2,
                   float T3, float* rel, float* sup, float* conn, float* cand_conf, int* cand_pred, const unsigned char* iou_mask, void* stream);

/* Per-image stable descending top-K over the accumulated candidates   (evaluator.py:292-316: confidence += connectivity, argsort, top 100).
 * conf [n_cand] f32 (already including the connectivity term), seg_ptr [n_img+1]; out_idx [n_img][K] image-local candidate
 * indices (-1 padded), out_count [n_img]. */
int sgc_topk_per_image(const float* conf, const int* seg_ptr, int n_img, int K, int* out_idx, int* out_count, void* stream);

/* Recall@K hit test (evaluator.py:306-356; Evaluator_Top3 :720-760 with n_pred = 3): for each of n_targets connected ground-truth
 * triples (t_row = row of its image in keep_pos; t_rel / t_scat / t_ocat int64; boxes f32 (x0,x1,y0,y1), int() truncation and slice
 * clipping applied inside) the rank of the first of the image's ranked candidates (keep_pos [n_img][K] flat positions, keep_cnt
 * [n_img]) whose labels match (equiv: optional n_equiv x n_equiv u8 equivalence table of utils.compare_object_cat, SGDET/SGCLS),
 * both rasterised-grid IoUs are >= iou_thresh and one of its n_pred predicates equals t_rel; K when there is none. */
int sgc_recall_hits(const long* c_scat, const long* c_ocat, const long* c_pred, int n_pred, const float* c_sbox, const float* c_obox,
                    const int* keep_pos, const int* keep_cnt, int K, const int* t_row, const long* t_rel, const long* t_scat,
                    const long* t_ocat, const float* t_sbox, const float* t_obox, int n_targets, const unsigned char* equiv, int n_equiv,
                    int feature_size, double iou_thresh, int* hit, void* stream);

/* "iou_mask" of testing(): the two boxes overlap on the FxF grid   (train_test.py:403-408).  bbox [n_obj][4], out u8 [n_pairs]. */
int sgc_overlap_filter(const int* bbox, const int* sub_idx, const int* obj_idx, unsigned char* out, int n_pairs, void* stream);

/* Commonsense filter of eval_cs/train_cs (evaluator.py:189-194,261-266: `tuple(...) in commonsense_violated_triplets` /
 * `not in commonsense_aligned_triplets` per candidate -> confidence = -inf): bit lookups in C*R*C-bit device bitmaps
 * (bit index (sub*R + rel)*C + obj).  scat/pred/ocat int64 [n], conf f32 [n] updated in place. */
int sgc_commonsense_filter(const long* scat, const long* pred, const long* ocat, float* conf, int n, const unsigned* aligned,
                           const unsigned* violated, int C, int R, void* stream);

/* train_cs flags per candidate (train_utils.py:49-50): weak = triplet not in the aligned set, strong = in the violated set (f32 0/1). */
int sgc_commonsense_flags(const long* scat, const long* ocat, const int* cand_pred, int n_pairs, int n_cand, const unsigned* aligned,
                          const unsigned* violated, int C, int R, float* weak, float* strong, void* stream);

/* ----------------------------------------------------------------------------------------------- pair tables (row a2) */

/* All ordered pairs of a minibatch in the reference's call order, built ON THE DEVICE from O(images + max objects) integers
 * (train_test.py:174-258 = :373-437, evaluate.py:132-183: `for graph_iter: keep_in_batch = nonzero(num > graph_iter); for edge_iter
 * < graph_iter: direction 1, direction 2`).  n_per_img [n_img], img_ptr [n_img+1] object ranges, goff [max_n+1] first pair of
 * every graph_iter block (goff[g+1] = goff[g] + 2*g*#{images with more than g objects}); n_pairs = goff[max_n].
 * Outputs (all int32): sub_idx / obj_idx / step (direction-step ordinal) / image [n_pairs]; pid [n_obj][pid_ld] (subject object,
 * object index inside the image) -> pair index or -1; obj_ptr [n_obj+1] + sub_list / obj_list [n_pairs]: the pairs of every
 * object as subject / as object in ascending pair order (CSR of the pair contraction); obj_img [n_obj]; step_ptr [T+1] with
 * T = max_n*(max_n-1).  rel_tri / dir_tri (may both be NULL): per image cat(relationships[b]) / cat(subj_or_obj[b])
 * (n(n-1)/2 entries, images concatenated; train_test.py:174-180); then raw [n_pairs] = the pair's stored predicate and
 * directed [n_pairs] = it where the stored direction equals the pair's direction, else -1 (train_utils.py:169-187). */
int sgc_scene_tables(const int* n_per_img, const int* img_ptr, const int* goff, int n_img, int max_n, int n_pairs, int n_obj, int pid_ld,
                     const int* rel_tri, const float* dir_tri, int* sub_idx, int* obj_idx, int* step, int* image, int* directed,
                     int* raw, int* pid, int* obj_ptr, int* sub_list, int* obj_list, int* obj_img, int* step_ptr, void* stream);

/* Stable bucket placement of a window list - the row plan of conv3 / fc1 over shared windows (no reference counterpart: the reference
 * computes every window of every pair, model.py:145-148).  codes[e] = pair*64 + window in pair order; key = window (img_key 0) or
 * obj_img[sub_idx[pair]]*64 + window (img_key 1).  mode 0: out[e] = base[key] + (number of earlier entries with that key);
 * mode 1: out = the entry indices ordered by key, stable, and seg [n_keys+1] = first position of every key.  Equals a stable sort by
 * key, bit for bit, without one. */
int sgc_bucket_place(const int* codes, int n, const int* sub_idx, const int* obj_img, int img_key, int n_keys, const int* base, int* out,
                     int* seg, int mode, void* stream);
/* the same placement through two-level kernels (a workgroup per (key, segment of 8192 entries): count, key offsets, place) - same ranks,
 * 0.09 -> 0.02 ms per call at the benchmark's sizes; scratch: int[sgc_bucket_place_scratch_ints(n, n_keys)] */
long sgc_bucket_place_scratch_ints(int n, int n_keys);
int sgc_bucket_place_seg(const int* codes, int n, const int* sub_idx, const int* obj_img, int img_key, int n_keys, const int* base, int* out,
                         int* seg, int mode, int* scratch, long scratch_ints, void* stream);

/* Pair lists of the pseudo-pairs of conv3 over shared windows (csrc/kernels_shared.hip): ps_sub / ps_obj [2 n_obj (+ n_img)] = (o, bg(o)),
 * (bg(o), o) (, (bg, bg) per image) with bg(o) = n_obj + obj_img[o]; bg_codes [64 n_img] (may be NULL) = window codes of the background
 * maps, pair index n_pairs + 2 n_obj + image; *bg_n = 64 n_img. */
int sgc_pseudo_pair_tables(const int* obj_img, int n_obj, int n_img, int n_pairs, int with_bg, int* ps_sub, int* ps_obj, int* bg_codes,
                           int* bg_n, void* stream);

/* out [rows][n] = inclusive prefix sums of every row of in (int32; in == out allowed): the entry counts of the window lists. */
int sgc_scan_rows(const int* in, int* out, int rows, int n, void* stream);

/* Rows of the other entries of the window list in the window-major row space: the per-object entries (codes = (P + pseudo-pair)*64 +
 * window: dest = goff[window] + pseudo-pair) and the entries of the CONV list (pairs that convolve their own windows; incl_* = inclusive
 * entry counts over the pair index space of the two lists): dest_conv[e] = dest_all[e - first_conv(pair) + first_all(pair)]. */
int sgc_window_rows_objects(const int* codes, int n, const int* goff, int n_pairs, int* dest, void* stream);
int sgc_window_rows_conv(const int* codes_conv, int n, const int* incl_conv, const int* incl_all, const int* dest_all, int* dest_conv,
                         void* stream);

/* Per-pair loss coefficients (train_utils.py:64-94,116-157; the running sums of train_test.py:219-258 give step t the weight T-t):
 * loss_i = -a*super[st] - b*rel[t] + c*BCE(conn, y); one thread per direction-step, double arithmetic in pair order.
 * class_weight [R] f32 (train_test.py:104-105); hier = 0: one class-weighted cross entropy (model.py:37-102 variant). */
int sgc_loss_coefficients(const int* step_ptr, int n_steps, const int* directed, const float* class_weight, int ng, int np, int hier,
                          double lambda_connectivity, double lambda_not_connected, int* tgt, float* coef_a, float* coef_b, float* coef_c,
                          float* conn_y, void* stream);

/* Connectivity statistics of train_one_direction / evaluate_one_direction summed over the minibatch (train_utils.py:66-87,176-184):
 * out5 = {not connected, connected, predicted connected (sigmoid >= 0.5), precision numerator (predicted connected and the stored
 * predicate of the unordered pair != -1), recall numerator (connected and round(sigmoid) == 1)}.  included (u8, may be NULL):
 * count only the pairs of the steps the overlap filter kept (testing()). */
int sgc_connectivity_stats(const float* conn, const int* directed, const int* raw, const unsigned char* included, int n_pairs,
                           unsigned long long* out5, void* stream);

/* Per-object label vectors (model.py:152-168 one-hot / multi-hot concat as additive rows of fc2): lsub[o] = W[:, col0+cat] +
 * sum_k mh[o][k] * W[:, col0+2C+k], lobj[o] = W[:, col0+C+cat] + sum_k mh[o][k] * W[:, col0+2C+S+k]; W = fc2.weight [512][ld] f32,
 * super_multihot [n_obj][S] f32 or NULL (OIV6).  sgc_label_grads is its transpose: writes columns col0 .. col0+2C+2S-1 of
 * grad_fc2_weight from dlsub / dlobj [n_obj][512] (deterministic object order); the subject-role and object-role rows may
 * carry different labels (the per-step call of model.py:170 labels its b subject crops and b object crops separately). */
int sgc_label_vectors(const float* fc2_weight, int ld, int col0, const long* cats, const float* super_multihot, int n_obj, int C, int S,
                      float* lsub, float* lobj, void* stream);
int sgc_label_grads(const float* dlsub, const float* dlobj, const long* cats_sub, const long* cats_obj, const float* mh_sub,
                    const float* mh_obj, int n_obj, int C, int S, float* grad_fc2_weight, int ld, int col0, void* stream);

/* ----------------------------------------------------------------------------------------------- backward */

/* Forward expansion for training: writes z in f16 (conv3 forward operand) and bf16 (conv3 weight-gradient operand) and records
 * the relu/maxpool routing (amz [n_pairs*256][256] bytes: two 4-bit codes per byte - channel 2k in the low nibble, 2k+1 in the high one;
 * code = winning position 0..3, 4 = none).  Any output may be NULL. */
int sgc_pair_expand_train(const void* U, const void* V, const int* sub_idx, const int* obj_idx, void* z_pad_f16, void* z_pad_bf16,
                          unsigned char* amz, int n_pairs, void* stream);

/* Loss + head backward per pair   (train_utils.py:64-94,116-157, utils.py:28-35 with the step weights of train_test.py:219-258 folded
 * into per-pair coefficients by the host): loss_i = -a*super[st] - b*rel[t] + c*BCE(conn, y).
 * W [64][512] f32 head rows; dl [n_pairs][64] dL/dlogits; loss [n_pairs]; dpre [n_pairs][512] bf16 = dL/d(fc2 pre-activation);
 * dp_extra (may be NULL) [n_pairs][512] f32 is added to dL/d(hidden) before the ReLU/dropout mask (contrastive term);
 * cs_coef (may be NULL) [n_pairs][3|1] adds the train_cs penalty cs_coef * max softmax per candidate (train_utils.py:36-62),
 * using the forward candidates cand_conf / cand_pred. */
int sgc_head_loss_bwd(const float* rel, const float* sup, const float* conn, const float* p, const int* tgt, const float* coef_a,
                      const float* coef_b, const float* coef_c, const float* conn_y, const float* W, int n_pairs, int ng, int np, int ns,
                      int hier, float T1, float T2, float T3, float drop_scale, float* dl, float* loss, void* dpre, const float* dp_extra,
                      const float* cs_coef, const float* cand_conf, const int* cand_pred, void* stream);
/* Head backward for ARBITRARY upstream gradients (the per-step `relation_classifier(...)` call of train_utils.py:26-27 under the
 * caller's autograd, `losses.backward()` train_test.py:276): g_rel [n_pairs][R] = dL/d(relation outputs: log-probs, or raw logits
 * when hier = 0), g_sup [n_pairs][3] (may be NULL), g_conn [n_pairs] (may be NULL), g_hidden [n_pairs][512] = dL/d(hidden output)
 * (may be NULL).  Writes dl [n_pairs][64] = dL/d(head logits) and dpre [n_pairs][512] bf16 = dL/d(fc2 pre-activation). */
int sgc_head_bwd_upstream(const float* rel, const float* sup, const float* p, const float* g_rel, const float* g_sup, const float* g_conn,
                          const float* g_hidden, const float* W, int n_pairs, int ng, int np, int ns, int hier, float T1, float T2,
                          float T3, float drop_scale, float* dl, void* dpre, void* stream);
/* part [ceil(n_pairs/chunk)][64][513] f32 partial head weight (cols 0..511) and bias (col 512) gradients */
int sgc_head_wgrad(const float* dl, const float* p, float* part, int n_pairs, int chunk, void* stream);

/* Supervised-contrastive loss of the training step (sup_contrast/losses.py:85-181 SupConLossHierar on the hidden vectors of
 * the connected pairs and of their augmented view; train_test.py:260-273): loss_rows [2M], G [2M][2M] scratch, dF [2M][512]. */
int sgc_supcon_hierar(const float* F, const int* labels, int M, float temperature, float grad_scale, float* G, float* loss_rows,
                      float* dF, void* stream);

int sgc_slab_sum(const float* in, float* out, long n, int slabs, int accumulate, void* stream);          /* out[n] (+)= sum_s in[s][n] */
int sgc_fill_zero(void* ptr, long nbytes, void* stream);   /* 16-byte aligned ptr; workspace creation (zero halos are written once) */
int sgc_slab_sum_ld(const float* in, float* out, int rows, int cols, long ld_out, int slabs, void* stream);  /* out[r*ld_out+c] = sum_s in[s][r][c] */
int sgc_colsum(int elem, const void* X, float* part, long rows, int cols, int row_blocks, void* stream);  /* bias gradients */
int sgc_segment_sum_rows(const void* X, const int* ptr, const int* list, float* out, int n_seg, int cols, void* stream); /* label-column grads */
int sgc_convert_f16_bf16(const void* in, void* out, long n, void* stream);
/* 64x64 tile transposes with conversion (weight-copy derivation / gradient layout): dst[a*sa_d + b*sb_d + j*ds_j + i] =
 * convert(src[a*sa_s + b*sb_s + i*ss_i + j]) for a < na, b < nb, i,j < 64; out_kind 0 f16, 1 bf16, 2 f32. */
int sgc_transpose_cast(const float* src, void* dst, int out_kind, int na, int nb, long sa_s, long sb_s, long ss_i, long sa_d,
                       long sb_d, long ds_j, void* stream);
/* Weight layouts in one launch each (no reference counterpart: the reference's conv / linear ops read the f32 parameters directly;
 * the 16-bit compute copies here are re-derived after every optimizer step).  dst[dst_off + sum i_k*dst_strides[k]] =
 * cast(src[src_off + sum i_k*src_strides[k]]) over the ndim <= 5 dimensions dims (the last is the fastest; strides in elements, any
 * sign - a flipped 3x3 kernel is stride -1 from offset 8); out_kind 0 f16, 1 bf16, 2 f32.  src / dst host arrays are read at call time. */
int sgc_permute_cast(const float* src, void* dst, int out_kind, int ndim, const int* dims, const long* src_strides, const long* dst_strides,
                     long src_off, long dst_off, void* stream);
/* n_seg <= 48 [rows][cols] blocks of one source tensor (shared row / column strides), block s to dst_off[s] with row pitch dst_ld[s]
 * from src_off[s]: the stacked tap matrices of the patch form of the conv3 data gradient (sgc_windows_dgrad_patches). */
int sgc_segment_cast(const float* src, void* dst, int out_kind, int rows, int cols, long src_row_stride, long src_col_stride, int n_seg,
                     const long* dst_off, const long* dst_ld, const long* src_off, void* stream);

int sgc_fc2_dgrad(const void* dpre, const void* w2mT, const void* h1, void* dh1, int n_pairs, float drop_scale, void* stream);
/* Weight gradients with a `splits` argument write split-K partial sums: slabs [*n_slabs][M][N] f32, reduced by sgc_slab_sum.
 * splits <= 0 lets the library choose (whole waves of blocks on the 256 CUs, >= 8 K tiles per block): at most 7 slabs for
 * sgc_conv3_wgrad and 25 for sgc_conv2_wgrad. */
int sgc_fc2_wgrad(const void* dpre, const void* h1_bf16, float* slabs, int n_rows, int splits, int* n_slabs, void* stream);
int sgc_fc1_dgrad(const void* dh1, const void* w1pT, void* dy, int n_pairs, int K, void* stream);
int sgc_fc1_wgrad(const void* dh1, const void* y_bf16, float* dw, int n_rows, int K, void* stream);
int sgc_unpool_relu_bwd(const void* dy, const unsigned char* argmax, void* dy3_pad, float* dbias_part, int* n_parts, int n_pairs, void* stream);
int sgc_conv3_dgrad(const void* dy3_pad, const void* wd3, void* dz, int n_pairs, void* stream);
/* The same data gradient with the un-pool fused into the operand staging: dy [n_pairs*64][1024] bf16 = gradient of the POOLED conv3
 * output (what sgc_fc1_dgrad writes), argmax = the routing byte of sgc_conv3_relu_pool (model.py:145-146 backward: ReLU + 2x2 max-pool
 * route the gradient to one pixel per window and channel); no un-pooled tensor is materialised. */
int sgc_conv3_dgrad_pooled(const void* dy, const unsigned char* argmax, const void* wd3, void* dz, int n_pairs, void* stream);
int sgc_conv3_wgrad(const void* dy3_pad, const void* z_pad_bf16, float* slabs, int n_pairs, int splits, int* n_slabs, void* stream);
/* The same weight gradient on the sparse matrix cores (2:4 structured sparsity: at most one non-zero per 2x2 pooling window):
 * dy [n_pairs*64][1024] bf16 = the POOLED gradient (output of sgc_fc1_dgrad), argmax = the routing byte of sgc_conv3_relu_pool;
 * pack_ac (n_pairs*4*1024*64 bytes) and pack_ic (n_pairs*4*1024*8 bytes) are scratch for the packed operand. */
int sgc_conv3_wgrad_sparse(const void* dy, const unsigned char* argmax, const void* z_pad_bf16, void* pack_ac, void* pack_ic,
                           float* slabs, int n_pairs, int splits, int* n_slabs, void* stream);
/* sgc_unpool_relu_bwd that also writes the packed operand in the same pass (dbias_part: at most 768 partials; dy3_pad may be NULL
 * when the data gradient un-pools on the fly, sgc_conv3_dgrad_pooled);
 * sgc_conv3_wgrad_sparse called afterwards with dy == NULL uses pack_ac / pack_ic as they are. */
int sgc_unpool_relu_bwd_pack(const void* dy, const unsigned char* argmax, void* dy3_pad, float* dbias_part, int* n_parts,
                             void* pack_ac, void* pack_ic, int n_pairs, void* stream);
/* dU_pad [n_obj][34][34][512] bf16 = sum over the pairs listed for each object of the routed dz  (transpose of the expansion);
 * amz: the nibble-packed routing codes written by the expansion. */
int sgc_pair_contract(const void* dz, const unsigned char* amz, const int* ptr, const int* list, void* dU_pad, int n_obj, void* stream);
int sgc_conv2_dgrad(const void* dU_pad, const void* wd2, void* da, int n_obj, void* stream);
/* The same on the objects' GRADIENT regions (the backward of model.py:141-143 restricted to where the gradient can be non-zero): the pair
 * contraction writes dU_o only inside the pixel rectangle of o's pseudo-pair, so the 2x2-pixel cells outside it hold exact zeros.
 * sgc_conv2_bwd_regions lists (object * 256 + cell) for the n_real boxed objects (their rectangle widened by ``dilate`` cells: 1 for the
 * data gradient) and all cells of the n_objx - n_real background objects behind them; sgc_conv2_dgrad_regions
 * writes the rows of the listed cells of da (the caller zero-fills the rest: bit-identical to sgc_conv2_dgrad). */
int sgc_conv2_bwd_regions(const int* bbox, int n_real, int n_objx, int dilate, int* gather, int* n_out, void* stream);
int sgc_conv2_dgrad_regions(const void* dU_pad, const void* wd2, const int* gather, const int* gather_n, int max_entries, void* da,
                            void* stream);
int sgc_conv2_wgrad(const void* dU_pad, const void* a_pad_bf16, float* slabs, int n_obj, int splits, int* n_slabs, void* stream);
/* dA [n_img*F*F][D] f32 = gradient of the per-image map (inside the boxes); dcst_part [*n_parts][D] f32 = partial sums of the gradient
 * of the tanh(b1) constant (outside the boxes), *n_parts = n_img * F*F*D/2048 rows, reduced by sgc_slab_sum (fixed order, no atomics). */
int sgc_object_masked_maps_bwd(const void* da, const int* img_ptr, const int* bbox, float* dA, float* dcst_part, int* n_parts, int n_img,
                               int F, int D, void* stream);
int sgc_tanh_bwd(const float* dA, const void* a_img, void* dpre, long n, void* stream);
int sgc_conv1_wgrad(const void* dpre, const void* x_bf16, float* slabs, int n_rows, int XC, int splits, int* n_slabs, void* stream);

/* One-pass SGD with momentum and weight decay on one f32 parameter tensor (torch.optim.SGD semantics with dampening 0, no Nesterov:
 * the optimizer of train_test.py:99-100): g' = g + wd*w; buf = first_step ? g' : momentum*buf + g'; w -= lr*buf.
 * 16-byte aligned pointers take the float4 path, anything else a scalar one. */
int sgc_sgd_momentum_step(float* w, const float* g, float* momentum_buf, long n, float lr, float momentum, float weight_decay,
                          int first_step, void* stream);
/* The same update for fc1.weight ([rows][1024 channels * 64 windows] f32, reference order channel*64 + window: model.py:118) with the
 * gradient given in GEMM order ([rows][window*1024 + channel], what sgc_fc1_windows_wgrad / sgc_fc1_wgrad write) - the transposition
 * of the gradient back to the reference order, the update (train_test.py:100,277) and the f16 compute copy of the forward
 * (w1p_f16 [rows][window*1024 + channel], may be NULL) in ONE pass over gradient, weight and momentum buffer.  rows may be a row range
 * (pointers at its first row).  Weights and momentum bit-identical to sgc_transpose_cast(kind 2) + sgc_sgd_momentum_step, the copy to
 * sgc_transpose_cast(kind 0) of the result. */
int sgc_sgd_fc1_fused(float* w, const float* g_gemm_order, float* momentum_buf, int rows, float lr, float momentum, float weight_decay,
                      int first_step, void* w1p_f16, void* stream);
/* The same update for up to 32 SMALL tensors in one launch (w / g / momentum_buf / n: host arrays of n_tensors device pointers / element
 * counts, read at call time; bit t of first_mask = tensor t has no momentum buffer yet).  18 of the head's 22 parameter tensors are
 * biases, head rows and 1x1 convolutions: one launch each cost more in launch gaps than in work. */
int sgc_sgd_momentum_multi(int n_tensors, float* const* w, const float* const* g, float* const* momentum_buf, const long* n, float lr,
                           float momentum, float weight_decay, unsigned first_mask, void* stream);

/* ----------------------------------------------------------------------------------------------- SGDET / SGCLS object front-end
 * (evaluate.py:309-366 = :543-589, utils.py:58-74,377-425)
 *
 * sgc_detr_candidates: per (image, query) softmax over the C1 decoder logits, has-object test (arg-max < num_classes), top-k
 * probabilities/classes, DETR (alphabetical) -> dataset (frequency) class index through alp2fre [C1], box cxcywh in [0,1] ->
 * (x0,x1,y0,y1) * feature_size with clamp   (evaluate.py:311-332).  cand_cat [n_img][n_query][topk] holds the mapped class or -1
 * when the query has no object or the mapped class equals num_classes (:323,340-344); cand_conf the probabilities;
 * cand_box [n_img][n_query][4]. */
int sgc_detr_candidates(const float* logits, const float* boxes, const int* alp2fre, int n_img, int n_query, int C1, int num_classes,
                        int topk, float feature_size, int* cand_cat, float* cand_conf, float* cand_box, void* stream);
/* Per-class greedy NMS of every image's n_query*topk candidate slots (evaluate.py:347-366; torchvision.ops.nms 0.15.2: stable
 * descending score order, suppress when inter/(a_i + a_j - inter) > iou_threshold).  out_slot [n_img][n_query*topk]: kept slot
 * indices (slot = query*topk + rank) in the reference's concatenation order (classes ascending, scores descending), -1 padded;
 * out_count [n_img].  n_query*topk <= 512. */
int sgc_nms_per_class(const int* cand_cat, const float* cand_conf, const float* cand_box, int n_img, int n_query, int topk,
                      double iou_threshold, int* out_slot, int* out_count, void* stream);
/* SGCLS label matching (utils.py:377-425): for every ground-truth box [tgt_ptr segments] the two predicted boxes
 * [pred_ptr segments] of its image with the largest rasterised-grid IoU (utils.py:58-74); equal IoUs resolve to the lower
 * prediction index.  top_idx / top_iou [n_tgt][2] (image-local prediction indices, -1 / -1.0 when the image has < 2 predictions). */
int sgc_match_boxes_top2(const float* pred_box, const int* pred_ptr, const float* tgt_box, const int* tgt_ptr, int n_img, int max_tgt,
                         int feature_size, int* top_idx, float* top_iou, void* stream);

/* ----------------------------------------------------------------------------------------------- generic-size trunk
 * model.py:110-111 constructs BayesianRelationClassifier(args, input_dim, feature_size, ...) with ANY sizes; the tiled kernels above are
 * specialised to input_dim = 128, feature_size = 32 (every shipped configuration, main.py:49-85).  These entry points run the trunk
 * (model.py:138-150) for every OTHER size in plain f32, per pair, literally (csrc/kernels_generic.hip); fc2 / head / loss and their
 * backward are size-independent and stay on the kernels above.  C = input_dim, F = feature_size (a multiple of 4).  All tensors f32
 * channels-last unless stated; img / box: [2][n_pairs] image index and [2][n_pairs][4] slice-normalised (x0,x1,y0,y1) of the subject
 * (side 0) and object (side 1) of every pair; feat / depth: NCHW with per-image element strides (the minibatch's tensors, or the
 * pre-masked [b,2C+1,F,F] crops of forward() with depth = feat + 2C*F*F). */

/* a [n_pairs][F*F][2C] = cat(tanh(conv1_1(feat*mask_s, depth*mask_s)), tanh(conv1_2(... mask_o)))   (train_test.py:164-169,194-195,
 * model.py:139-141).  w1 [2][C][2C+1], b1 [2][C]: conv1_1 / conv1_2 in the reference's layout. */
int sgc_generic_conv1_tanh(const float* feat, const float* depth, long stride_feat, long stride_depth, const int* img, const int* box,
                           const float* w1, const float* b1, int n_pairs, int C, int F, float* a, void* stream);
/* out [n_pairs][(S/2)^2][Cout] = maxpool2(relu(conv3x3(in [n_pairs][S*S][Cin], w [Cout][Cin][3][3], padding 1) + b)); code: one byte per
 * output, dy*2+dx of the first maximum, 4 = no positive value   (model.py:142-144 conv2_1, :145-147 conv3_1). */
int sgc_generic_conv3x3_relu_pool(const float* in, const float* w, const float* b, int n_pairs, int S, int Cin, int Cout, float* out,
                                  unsigned char* code, void* stream);
/* Backward of the above from the pooled gradient dout: dpre [n_pairs][S*S][Cout] scratch (un-pooled), din (may be NULL), dw [Cout][Cin][3][3],
 * db [Cout]; fixed summation order, no atomics. */
int sgc_generic_conv3x3_bwd(const float* in, const float* w, const float* dout, const unsigned char* code, int n_pairs, int S, int Cin, int Cout,
                            float* dpre, float* din, float* dw, float* db, void* stream);
/* h1 [n_pairs][4096] f16 = dropout(relu(fc1(flatten_NCHW(y))))   (model.py:148-149; y [n_pairs][Q][C8], Q = (F/4)^2, C8 = 8C; w1 [4096][C8*Q] in
 * the reference's column order c*Q + q; dropout keep bit = the hash of common.h:dropout_keep over p*4096 + n, x2). */
int sgc_generic_fc1_relu(const float* y, const float* w1, const float* b1, int n_pairs, int Q, int C8, int dropout, unsigned seed, void* h1,
                         void* stream);
/* dh1 [n_pairs][4096] bf16 (gradient wrt fc1's pre-activation, as sgc_fc2_dgrad leaves it) -> dy, dw1 [4096][C8*Q], db1 [4096]. */
int sgc_generic_fc1_bwd(const void* dh1, const float* y, const float* w1, int n_pairs, int Q, int C8, float* dy, float* dw1, float* db1,
                        void* stream);
/* da (gradient wrt a, overwritten with the gradient wrt conv1's pre-activation) -> dw1 [2][C][2C+1], db1 [2][C]. */
int sgc_generic_conv1_bwd(const float* feat, const float* depth, long stride_feat, long stride_depth, const int* img, const int* box, const float* a,
                          float* da, int n_pairs, int C, int F, float* dw1, float* db1, void* stream);

/* ----------------------------------------------------------------------------------------------- test hooks (raw GEMM engines) */
int sgc_dbg_gemm_nt(int elem, const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc, const float* bias, void* stream);
int sgc_dbg_gemm_nt_abl(int abl, const void* A, const void* B, void* C, int M, int N, int K, void* stream);
/* tools/stride_microbench.py: the ping-pong NT block with the operands' row pitches as parameters (power-of-two pitches cost nothing: profiles/r05_stride_microbench.txt) */
int sgc_dbg_gemm_nt_ld(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, void* stream);
int sgc_dbg_fc1_windows_gemm(const void* ywm, const void* w, const int* tile_group, void* owm, int rows, long ldb, long group_stride, long ldc, int mode, int stagger, int phases, unsigned long long* clk, void* stream);
int sgc_dbg_dgrad_patches(const void* dy3x, const void* w3patch, void* patch, int entries, long lda, long seg_stride, int bpad, int split, void* stream);
int sgc_dbg_conv_nt(int elem, const void* A, const void* B, void* C, int n_img, int lgS, int Cin, int N, const float* bias, void* stream);
int sgc_dbg_gemm_tn(int elem, const void* A, const void* B, float* C, int M, int N, int K, long lda, long ldb, int splits, int* slabs, void* stream);
int sgc_dbg_conv_tn(int elem, const void* A, const void* B, float* C, int M, int n_img, int lgS, int Cin, int splits, int* slabs, void* stream);
int sgc_dbg_tr_probe(const int* addr, short* out, void* stream);
int sgc_dbg_mfma_rate(float* out, int blocks, int iters, int mode, void* stream);
int sgc_dbg_smfmac_probe(const void* a, const void* b, const int* idx, float* c, int abid, void* stream);

#ifdef __cplusplus
}
#endif
#endif
