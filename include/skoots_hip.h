/* skoots_hip.h -- C ABI of libskoots_hip.so (MI355X / gfx950).
 *
 * The reference (buswinka/skoots) has no FFI: its hot path sits behind the Python
 * function skoots.lib.eval.eval() and the library functions it calls.  Each entry
 * point below names the reference code it replaces (file:line, relative to the
 * reference tree).  The Python host mirror of the reference interface lives in
 * skoots_amd/lib/ and binds these symbols with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - volumes are C-contiguous (X, Y, Z) with Z fastest, exactly the reference's
 *     (C, X, Y, Z) tensors per channel;
 *   - `stream` is a hipStream_t passed as void*; all work is stream-ordered, no
 *     entry point synchronises unless its comment says so;
 *   - no entry point allocates device memory: callers pass workspaces;
 *   - return value: SK_OK, or a negative code; sk_last_error() gives the text.
 */
#ifndef SKOOTS_HIP_H
#define SKOOTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SK_OK 0
#define SK_ERR_ARG (-1)
#define SK_ERR_HIP (-2)
#define SK_ERR_CAPACITY (-3)

#define SK_F16 0
#define SK_F32 1
#define SK_I16 2
#define SK_I32 3
#define SK_U8 4

const char* sk_last_error(void);
int sk_abi_version(void);

/* ------------------------------------------------------------------------ *
 * Stage 3: offset following + label assignment
 * ------------------------------------------------------------------------ */

/* Planar (3, X, Y, Z) fp16 vectors -> interleaved (X, Y, Z, 4) fp16 (4th lane 0).
 * Internal staging layout for the follow kernel: one 8-byte load per hop instead
 * of three cache lines.  No reference counterpart (layout choice). */
int sk_vec_interleave(const void* vec_planar, void* vec4, int64_t nvox, void* stream);

/* Inverse of sk_vec_interleave: (X,Y,Z,4) fp16 -> planar (3,X,Y,Z) fp16, the layout of
 * the reference's `vectors` zarr array (skoots/lib/eval.py:103). */
int sk_vec_deinterleave(const void* vec4, void* vec_planar, int64_t nvox, void* stream);

/* skoots.lib.vector_to_embedding.vector_to_embedding (vector_to_embedding.py:79-174)
 * on one crop: vec (3, w, h, d) planar, fp16 or fp32 (vec_dtype SK_F16/SK_F32);
 * embed (3, w, h, d) fp32.  step_scale_host[(n)*3]: row 0 = float(scale), row i>=1 =
 * float(decay^i) * float(scale) (fp32 product), prepared by the host mirror. */
int sk_vector_to_embedding(const void* vec, int vec_dtype, float* embed, int w, int h, int d,
                           const float* step_scale_host, int n_iter, void* stream);

/* skoots.lib.skeleton.index_skeleton_by_embed (skeleton.py:656-695):
 * out[i] = labels[round/clamp(embed[:, i])]; labels (LX, LY, LZ) int16 or int32,
 * embed (3, n) fp32, out (n) int32. */
int sk_index_skeleton_by_embed(const void* labels, int label_dtype, int lx, int ly, int lz,
                               const float* embed, int64_t n, int32_t* out, void* stream);

/* Fused stage 3 (eval.py:245-284): for every voxel of planes [z_lo, z_hi), find the
 * stage-3 crop that writes it last (owner tables built by the host from the
 * reference's crop generator, cropper.py:97-144), run the N-step follow inside that
 * crop's window with the reference's arithmetic, add the crop origin, gather the
 * label.  (X,Y,Z) is the GLOBAL volume; labels is the full (X,Y,Z) int16|int32 volume.
 * vec4 ((.,.,.,4) fp16) is an array over the z-window [win_lo, win_hi): shape
 * (X, Y, win_hi-win_lo, 4) -- the whole volume on one GPU, slab + halo when Z-sharded; the
 * window must contain every crop that owns a written plane.  out (int32) holds exactly
 * the written planes: shape (X, Y, z_hi-z_lo).  owner_{x,y,z}:
 * int32[X|Y|Z] origin of the owning crop per GLOBAL coordinate, or -1 (voxel stays 0). */
int sk_follow_assign(const void* vec4, const void* labels, int label_dtype, int32_t* out,
                     int X, int Y, int Z, int win_lo, int win_hi, const int32_t* owner_x,
                     const int32_t* owner_y, const int32_t* owner_z, int eff_w, int eff_h,
                     int eff_d, const float* step_scale_host, int n_iter, int z_lo, int z_hi,
                     void* stream);

/* ------------------------------------------------------------------------ *
 * Stage 1 tail: gate + dilate + threshold + interior scatter
 * ------------------------------------------------------------------------ */

/* eval.py:145-176 for a batch of n_tiles (<= 16) tiles in one launch.  out5 holds the 5-channel
 * network outputs, fp16 or fp32 (thresholds follow torch's scalar casting for that dtype); tile i
 * starts at element offset tile_offsets_host[i] and is addressed with the element strides
 * (stride_c, stride_x, stride_y, 1), extent (w, h, d) -- a (B,5,w,h,d) batch or views into a
 * larger (5,X,Y,Z) array alike.  For tile i the tile-local box [box_lo_host[3i..], box_hi_host[3i..])
 * (the tile interior, or the part of it this tile writes LAST in the reference's scatter
 * order, so that tiles can be scattered in any order / concurrently) is written into the
 * volume arrays at origins_host[3i..] + box: vec4 (X,Y,Z,4) fp16 and/or vec_planar (3,X,Y,Z)
 * fp16 (either may be NULL), skeleton (X,Y,Z) uint8 in {0,1}.  Dilation = 3x3x3 max then 3x3x1
 * max twice, zero padded at the tile faces (morphology.py:155-199). */
int sk_gate_dilate_scatter(const void* out5, int out_dtype, int n_tiles,
                           const int64_t* tile_offsets_host, int64_t stride_c, int64_t stride_x,
                           int64_t stride_y, int w, int h, int d, const int* origins_host,
                           const int* box_lo_host, const int* box_hi_host, void* vec4,
                           void* vec_planar, uint8_t* skeleton, int X, int Y, int Z,
                           float prob_thr, float skel_thr, void* stream);

/* skoots.lib.morphology.binary_dilation / binary_dilation_2d (morphology.py:155-199)
 * as library functions on a (w,h,d) fp32 map: max over a (2rx+1,2ry+1,2rz+1) window,
 * zero padded. */
int sk_max_filter3d(const float* in, float* out, int w, int h, int d, int rx, int ry, int rz,
                    void* stream);

/* ------------------------------------------------------------------------ *
 * Stage 2: skeleton labelling (skoots.lib.flood_fill.efficient_flood_fill,
 * flood_fill.py:13-122)
 * ------------------------------------------------------------------------ */

/* Bytes of workspace sk_ccl_crop needs for a crop of n voxels. */
size_t sk_ccl_workspace_bytes(int64_t crop_voxels);

/* flood_all (flood_fill.py:125-140) for one crop of the flood grid: 6-connected
 * labelling of src>0 inside the box [x0,x0+w) x [y0,y0+h) x [z0,z0+d) of the
 * (X,Y,Z) uint8 volume; components are numbered in raster order of their first
 * voxel (scipy.ndimage.label order) starting at state[0]+2, written to the int32
 * labels volume (same box).  state (device int32[4]): [0] running max id (init 1,
 * becomes 0 after an empty crop, exactly as the reference), [1] components in this
 * crop, [2] total components so far, [3] scratch. */
int sk_ccl_crop(const uint8_t* src, int32_t* labels, int X, int Y, int Z,
                int x0, int y0, int z0, int w, int h, int d,
                void* workspace, size_t workspace_bytes, int32_t* state, void* stream);

/* Seam scan (replaces get_adjacent_labels, flood_fill.py:237-261, with true
 * adjacency): for plane `v` of `axis` (0=x,1=y,2=z) emit the pairs
 * (labels[plane v], labels[plane v-1]) of face-adjacent nonzero voxels into
 * pairs (int32[2*capacity]); *count (device int32) is advanced atomically.
 * Duplicates are possible; the host sorts and uniques. */
int sk_seam_pairs(const int32_t* labels, int X, int Y, int Z, int axis, int v,
                  int32_t* pairs, int32_t* count, int capacity, void* stream);

/* HOST function (no GPU work): collision graph + depth-first components + "last id
 * represents the component" (flood_fill.py:82-105).  pairs_host: n pairs in the
 * reference's emission order.  Writes up to capacity (to_replace, replace_with)
 * entries; returns the number written or a negative code. */
int sk_seam_components_host(const int32_t* pairs_host, int n_pairs, int32_t* to_replace_host,
                            int32_t* replace_with_host, int capacity);

/* Device-side seam merge of the Z-sharded labelling (no reference counterpart: the reference is single device; the
 * partition is that of flood_fill.py:82-117).  Together they let a rank go from its local labelling to the merged global
 * ids without reading anything back to the host between the collectives:
 *   sk_compact_nonzero: flat positions of the non-zero labels into positions[capacity] (any order); *count (device,
 *     pre-zeroed) counts ALL of them, also past the capacity -- the caller detects an overflow from it later.
 *   sk_seam_union: meta (ranks, row_stride) int32 rows [components, pairs, -, - | pairs in rank-local ids ...] as
 *     all-gathered; offsets (ranks + 1) int64 exclusive prefix sums of the component counts (device); lut
 *     (lut_size, pre-filled with the identity) <- smallest global id of every id's component.  One workgroup.
 *   sk_relabel_lut_offset: labels[i] = lut[labels[i] + *offset] for labels[i] > 0 (offset: device scalar). */
int sk_compact_nonzero(const int32_t* labels, int64_t n, int64_t* positions, uint64_t* count, int64_t capacity, void* stream);
int sk_seam_union(const int32_t* meta, int ranks, int row_stride, int pair_capacity, const int64_t* offsets, int32_t* lut,
                  int64_t lut_size, void* stream);
int sk_relabel_lut_offset(int32_t* labels, int64_t n, const int32_t* lut, int64_t lut_size, const int64_t* offset, void* stream);

/* labels[i] = lut[labels[i]] for 0 <= labels[i] < lut_size (in place; flood_fill.py:177-234). */
int sk_relabel_lut(int32_t* labels, int64_t n, const int32_t* lut, int lut_size, void* stream);

/* ------------------------------------------------------------------------ *
 * Renumber (fastremap.renumber call, eval.py:304-306)
 * ------------------------------------------------------------------------ */

size_t sk_renumber_workspace_bytes(int64_t n, int max_label);

/* Distributed renumber, step 1: first[v] = min over this array of the GLOBAL C-order
 * index of label v (atomicMin into a table the caller pre-filled with 0xFFFFFFFF).
 * labels is the (X, Y, zl) slab at planes [z_off, z_off+zl) of a volume with Zg planes;
 * the ranks then all-reduce(MIN) the table, rank it and apply sk_relabel_lut. */
int sk_first_seen(const int32_t* labels, int X, int Y, int zl, int z_off, int Zg, int max_label,
                  uint32_t* first, void* stream);

/* Relabel to 1..K by first appearance in C order, 0 preserved, in place.
 * labels int32 (n), values in [0, max_label].  *n_labels (device int32) = K. */
int sk_renumber(int32_t* labels, int64_t n, int max_label, void* workspace,
                size_t workspace_bytes, int32_t* n_labels, void* stream);

/* ------------------------------------------------------------------------ *
 * Stage 1 body: U-Net layers (the network the reference builds with
 * cfg_to_bism_model, skoots/lib/utils.py:17-107, and runs at eval.py:117-143).
 * Graph and parity reference: oracle/unet_spec.py.  Activations are channels-last
 * fp16 (B, X, Y, Z, C).  A conv writes RAW outputs (bias added) plus per-block
 * GroupNorm partial sums; sk_groupnorm_finalize + sk_groupnorm_silu then normalise
 * and activate the tensor in place, so every conv input is already activated.
 * ------------------------------------------------------------------------ */

typedef struct sk_conv_src {
    const void* data;    /* (B, sx, sy, sz, c) fp16 channels-last                            */
    const float* affine; /* NULL: data is activated.  (B, 2, c) fp32: data is a RAW conv output, */
                         /* silu(a*x + b) is applied while it is staged / loaded              */
    int c;               /* channels of this source (multiple of 32 for ksize 3)            */
    int upsample;        /* 1: source is half resolution, read at (x>>1, y>>1, z>>1)        */
} sk_conv_src;

/* Implicit-GEMM 3-D convolution on the matrix cores (v_mfma_f32_32x32x16_f16).
 * ksize 3: stride 1, zero pad 1, input = channel concat of n_src (<= 2) sources, the
 * second optionally nearest-upsampled x2 (torch.cat([skip, interpolate(x)]) never
 * materialised).  ksize 2: stride 2, no pad.  ksize 1: pointwise.  weight: packed by
 * sk_conv3d_pack_weight_host; bias (cout) fp32; out (B, ox, oy, oz, cout) fp16 raw.
 * gn_partial: (B, sk_conv3d_num_blocks, cout/4, 2) fp32 per-block (sum, sumsq) of
 * the fp32 accumulators per channel quad, or NULL.  zero_page: 4 KiB; bytes [0, 1024) must
 * be zero and stay zero (source of the halo / padding lanes of the LDS-DMA), bytes
 * [2048, 3072) are write-only scratch for the ksize-3 kernel's masked store lanes.  Required for ksize 3 and for
 * the LDS-staged ksize-2 kernel ((cin, cout) = (32, 64) | (64, 128)).  A RAW source (affine != NULL) is activated
 * while it is staged / loaded: ksize 3 in LDS by the staging lanes, ksize 2 and 1 on load. */
int sk_conv3d(const sk_conv_src* srcs, int n_src, const void* weight, const float* bias,
              void* out, int B, int ox, int oy, int oz, int cout, int ksize,
              float* gn_partial, void* zero_page, void* stream);

/* sk_conv3d with a STORE BOX (round 3): store_box = {lo_x, lo_y, lo_z, hi_x, hi_y, hi_z} (host, tile-local, half open)
 * or NULL.  The convolution and its GroupNorm partial sums cover the whole tile as always; output voxels outside the
 * box may be left unwritten -- for a tensor whose only reader looks at a box of it (the last block's conv: the heads
 * evaluate the scatter's box, 28 % of a 300x300x20 tile).  Honoured by the COUT-32 kernels (conv3_px_kernel, and since
 * round 4 conv3_m16_kernel); wider layers store the whole tile, which satisfies the contract too. */
int sk_conv3d_box(const sk_conv_src* srcs, int n_src, const void* weight, const float* bias, void* out,
                  int B, int ox, int oy, int oz, int cout, int ksize, float* gn_partial,
                  void* zero_page, const int* store_box, void* stream);

/* sk_conv3d_box for the split precision (round 4; tensors and weight as sk_conv3d_split): the last block's conv of the
 * precision="split" network stores only the heads' box -- 28 % of the 14.7 GB a 64-tile batch of [hi | lo] lines is. */
int sk_conv3d_box_split(const sk_conv_src* srcs, int n_src, const void* weight, const float* bias, void* out,
                        int B, int ox, int oy, int oz, int cout, int ksize, float* gn_partial,
                        void* zero_page, const int* store_box, void* stream);

/* 2x2x2 stride-2 down conv with the GroupNorm + SiLU of its INPUT folded in (the fused form of north_star's
 * "fused GroupNorm+SiLU" for the two skip tensors): in_raw (B, 2ox, 2oy, 2oz, cin) fp16 is the RAW output of the
 * producing conv, affine (B, 2, cin) its GroupNorm coefficients.  The kernel stages the input through LDS, activates
 * it there (bit-identical to sk_groupnorm_silu), WRITES THE ACTIVATED VALUES BACK to in_raw -- afterwards the
 * tensor is activated, as its other consumer (the decoder's skip conv) needs it -- and computes the conv from LDS.
 * (cin, cout) = (32, 64) | (64, 128); weight / bias / out / gn_partial / zero_page as sk_conv3d with ksize 2
 * (which runs the same kernel without activation when its source has affine == NULL). */
int sk_conv3d_down_act(void* in_raw, const float* affine, const void* weight, const float* bias, void* out,
                       int B, int ox, int oy, int oz, int cin, int cout, float* gn_partial,
                       void* zero_page, void* stream);

/* sk_conv3d_down_act for the split precision (round 4): in_raw (B, 2ox, 2oy, 2oz, 2 cin) and out (B, ox, oy, oz, 2 cout)
 * hold [hi | lo] fp16 pairs per voxel line, weight = sk_conv3d_pack_weight_split_host(ksize 2).  Activates hi + lo with
 * the arithmetic of sk_groupnorm_silu_split (bit-identical), writes the pair back, three MFMA products per K step.
 * Same partial-sum rows as sk_conv3d_num_blocks(.., ksize 2) reports. */
int sk_conv3d_down_act_split(void* in_raw, const float* affine, const void* weight, const float* bias, void* out,
                             int B, int ox, int oy, int oz, int cin, int cout, float* gn_partial,
                             void* zero_page, void* stream);

/* sk_conv3d_down_act_split for precision "mix8": the same conv, but the activated input is written back as a mix8 line
 * [hi fp16 (cin) | per 32-channel chunk: x8 (32 bytes) | lo8 (32 bytes)] (sk_groupnorm_silu_mix8's format) -- for a skip
 * tensor whose other reader is sk_conv3d_upfold_mix8. */
int sk_conv3d_down_act_mix8(void* in_raw, const float* affine, const void* weight, const float* bias, void* out,
                            int B, int ox, int oy, int oz, int cin, int cout, float* gn_partial,
                            void* zero_page, void* stream);

/* Decoder conv over cat([skip, nearest-upsample x2 (up)]) with the upsample FOLDED INTO THE WEIGHTS of the upsampled
 * channels (csrc/conv3d_up.hip): 3x3x3, stride 1, zero pad 1, the same function as sk_conv3d(ksize 3) with sources
 * {skip, up (upsample = 1)} -- the first conv of each decoder level (oracle/unet_spec.py; skoots/lib/utils.py:17-107)
 * -- at 8 instead of 27 tap-chunks for the upsampled channels: an output voxel of parity p along an axis reads
 * U[x-1], U[x], U[x+1] = two distinct voxels of the low-resolution tensor, so per parity class (px, py, pz) the 27 taps
 * collapse to 2x2x2 taps whose weights are sums of the kernel's (formed on the host in fp32, rounded to fp16 once:
 * results agree with sk_conv3d to the fp16 rounding of those sums, exactly on integer-valued weights).
 * skip (B, ox, oy, oz, c_skip), up (B, ox/2, oy/2, oz/2, c_up): fp16, ACTIVATED; weight: sk_conv3d_pack_weight_upfold_host;
 * out (B, ox, oy, oz, cout) fp16 raw; gn_partial (B, sk_conv3d_upfold_num_blocks, cout/4, 2) or NULL.  cout = 32 | 64
 * (64: two launches of 32 output channels each).
 * sk_conv3d_upfold_num_blocks < 0: the geometry is not covered (an odd extent, oz/2 > 24) -- use sk_conv3d. */
int sk_conv3d_upfold(const void* skip, int c_skip, const void* up, int c_up, const void* weight,
                     const float* bias, void* out, int B, int ox, int oy, int oz, int cout,
                     float* gn_partial, void* stream);
int sk_conv3d_upfold_num_blocks(int ox, int oy, int oz, int cout);
/* HOST: torch-layout weight (cout, c_skip + c_up, 3, 3, 3) fp32 -> fragments of sk_conv3d_upfold.  Bytes needed / written. */
int64_t sk_conv3d_pack_weight_upfold_host(const float* w_host, int cout, int c_skip, int c_up, void* dst_host);
/* The same on split tensors (precision "split": every voxel line [hi (C) | lo (C)] fp16, value = hi + lo; out the same with
 * cout): three MFMA phases per logical chunk as sk_conv3d_split; the folded weights are summed in double and split afterwards. */
int sk_conv3d_upfold_split(const void* skip, int c_skip, const void* up, int c_up, const void* weight,
                           const float* bias, void* out, int B, int ox, int oy, int oz, int cout,
                           float* gn_partial, void* stream);
int64_t sk_conv3d_pack_weight_upfold_split_host(const float* w_host, int cout, int c_skip, int c_up, void* dst_host);
/* The same for precision "mix8": skip and up hold mix8 lines (sk_groupnorm_silu_mix8 / sk_conv3d_down_act_mix8), out is a RAW
 * split pair; per logical chunk one fp16 phase (hi halves x w_hi) and one block-scaled fp8 phase (K = 128 = two taps x
 * {w_lo . x8, w . lo8}; the folded weights are summed in double, then split into fp16 hi and the fp8 images).
 * weight + weight_scale_exp: sk_conv3d_pack_weight_upfold_mix8_host. */
int sk_conv3d_upfold_mix8(const void* skip, int c_skip, const void* up, int c_up, const void* weight, int weight_scale_exp,
                          const float* bias, void* out, int B, int ox, int oy, int oz, int cout,
                          float* gn_partial, void* stream);
int64_t sk_conv3d_pack_weight_upfold_mix8_host(const float* w_host, int cout, int c_skip, int c_up, void* dst_host,
                                               int* scale_exp);

/* Rows of gn_partial per batch item that sk_conv3d writes for this output shape. */
int sk_conv3d_num_blocks(int B, int ox, int oy, int oz, int cout, int ksize);

/* HOST: torch-layout weight (cout, cin, k, k, k) fp32 -> MFMA A-fragment order fp16.
 * Returns the bytes needed / written (dst_host == NULL only queries). */
int64_t sk_conv3d_pack_weight_host(const float* w_host, int cout, int cin, int ksize,
                                   void* dst_host);

/* One-pass stem (round 4): sk_conv3d_stem's normalisation + the conv ONCE, storing the RAW fp16 result (B, Xt, Yt, Zt, 32)
 * next to its GroupNorm partial sums -- for a consumer that activates a raw source itself (sk_conv3d with
 * sk_conv_src.affine: the single-chunk 32 -> 32 conv in LDS).  HipUNet.stem_single_pass selects it; the default stays the
 * two-pass form (statistics, then apply), which measured the same time end to end. */
int sk_conv3d_stem_raw(const void* image, int X, int Y, int Z, const int32_t* origins_host, int B, int Xt,
                       int Yt, int Zt, float mean, float stdv, const float* weight, const float* bias,
                       int cout, void* out_raw, float* gn_partial, void* workspace, size_t workspace_bytes,
                       void* stream);

/* Stem: first conv of the network (Cin = 1), fused with its GroupNorm + SiLU by running the
 * (cheap) conv twice instead of writing a raw tensor and re-reading it.
 * sk_conv3d_stem: cuts B tiles of extent (Xt,Yt,Zt) at origins_host[3*b..] out of the (X,Y,Z)
 *   fp16 image volume, normalises (x - mean)/std in fp16 arithmetic exactly as eval.py:139 into
 *   the zero-framed workspace, runs the conv (weights split into fp16 hi + lo, exact products,
 *   fp32 accumulation) and writes ONLY the GroupNorm partials (B, stem_num_blocks, 8, 2).
 * sk_conv3d_stem_apply (after sk_groupnorm_finalize): recomputes the conv from the same
 *   workspace, applies the affine (B, 2, 32) + SiLU and stores out (B, Xt, Yt, Zt, 32) fp16
 *   ACTIVATED.  weight (27, 32) fp32 [tap=(dx*3+dy)*3+dz][cout]. */
int sk_conv3d_stem(const void* image, int X, int Y, int Z, const int32_t* origins_host, int B,
                   int Xt, int Yt, int Zt, float mean, float std, const float* weight,
                   const float* bias, int cout, float* gn_partial, void* workspace,
                   size_t workspace_bytes, void* stream);
int sk_conv3d_stem_apply(int B, int Xt, int Yt, int Zt, const float* weight, const float* bias,
                         const float* affine, void* out, int cout, const void* workspace,
                         void* stream);
int sk_conv3d_stem_num_blocks(int X, int Y, int Z);
size_t sk_conv3d_stem_workspace_bytes(int B, int Xt, int Yt, int Zt);

/* GroupNorm statistics -> per-channel affine, reduced in a fixed order (deterministic):
 * gn_partial (B, nblocks, C/4, 2); affine (B, 2, C): a = gamma*rstd, b = beta - mean*a.
 * With more than 4096 rows per sample gn_partial is first compacted IN PLACE (a second kernel level instead of
 * one block reading every row); it is consumed by this call either way. */
int sk_groupnorm_finalize(float* gn_partial, int B, int nblocks, int groups, int C,
                          int64_t voxels, const float* gamma, const float* beta, float eps,
                          float* affine, void* stream);

/* Fused GroupNorm affine + SiLU, in place on (B, voxels, C) fp16. */
int sk_groupnorm_silu(void* x, const float* affine, int B, int64_t voxels, int C, void* stream);

/* Heads: 1x1x1 conv C->5, tanh on [0:3], sigmoid on [3:5]; out5 (B, 5, X, Y, Z) planar fp16 =
 * the reference's output layout (eval.py:145-147).  x: (B, X, Y, Z, C) fp16, activated if
 * affine == NULL, else RAW with affine (B, 2, C) applied (+ SiLU) on load.  weight (5, C) fp32,
 * bias (5).  box_lo/hi_host (int[3], tile-local, may be NULL = whole tile): only that box of every
 * tile is evaluated and written -- the eval pipeline needs just the interior + dilation reach. */
int sk_heads(const void* x, const float* affine, const float* weight, const float* bias, void* out5,
             int B, int X, int Y, int Z, int C, const int* box_lo_host, const int* box_hi_host,
             void* stream);

/* "split" precision mode: every activation is a pair of fp16 tensors hi + lo interleaved per voxel,
 * (B, x, y, z, 2C) = [hi (C) | lo (C)], value = hi + lo (~22 significant bits); weights are split the same way.
 * A product runs as three fp16 MFMAs with fp32 accumulation: w_lo*x_hi + w_hi*x_hi + w_hi*x_lo (the dropped
 * lo*lo term is ~2^-22 relative).  Meets the max-abs 1e-3 tolerance BASELINE.json's north_star states against
 * the fp32 oracle (the plain fp16-operand path sits at 4-7e-3, the reference's own fp16 autocast at eval.py:142
 * likewise) at ~3x the matrix-core work of the fp16 path instead of the 11x of the exact-fp32 instruction.
 * Same semantics as the entry points without the suffix; `c` of a source is its LOGICAL channel count,
 * sources must be activated (affine == NULL). */
int sk_conv3d_split(const sk_conv_src* srcs, int n_src, const void* weight, const float* bias,
                    void* out, int B, int ox, int oy, int oz, int cout, int ksize,
                    float* gn_partial, void* zero_page, void* stream);
/* HOST: torch-layout fp32 weight -> fragments for sk_conv3d_split (ksize 3: [lo, hi, hi] chunk triples of an
 * expanded 3*cin contraction; ksize 1 / 2: hi fragments then lo fragments).  Bytes needed / written. */
int64_t sk_conv3d_pack_weight_split_host(const float* w_host, int cout, int cin, int ksize,
                                         void* dst_host);
/* Precision "mix8" (round 4), 3x3x3, C -> C with C = 32 | 64 | 128: the split conv's two correction products (w_lo x, w x_lo) as
 * ONE block-scaled fp8 matrix product next to the fp16 product w_hi x_hi -- C = 32: v_mfma_scale_f32_16x16x128_f8f6f4, K = 128 =
 * two tap rows x {w_lo . x8, w . lo8} (conv3_m16_kernel); C = 64 | 128: v_mfma_scale_f32_32x32x64_f8f6f4, K = 64 = one tap x
 * {w_lo . x8, w . lo8} per 32-channel chunk (conv3_kernel) -- 2 to 2.1 instead of 3 MFMA passes per tap.
 * src: ONE activated source of 2 C-half lines [hi fp16 (C) | per 32-channel chunk: x8 = e4m3(16 x) (32 bytes) | lo8 =
 * e4m3(2^15 (x - hi)) (32 bytes)] (written by sk_conv3d_stem_apply_mix8 / sk_groupnorm_silu_mix8); weight + weight_scale_exp from
 * sk_conv3d_pack_weight_mix8_host; out / gn_partial / zero_page / store_box as sk_conv3d_box_split (out is a RAW split pair
 * [hi | lo]).  Error of the corrections: 2^-4 relative on terms that are 2^-11 of the result (tools/fp8_correction_study.py). */
int sk_conv3d_mix8(const sk_conv_src* srcs, int n_src, const void* weight, int weight_scale_exp, const float* bias,
                   void* out, int B, int ox, int oy, int oz, int cout, float* gn_partial, void* zero_page,
                   const int* store_box, void* stream);
/* HOST: torch-layout fp32 weight (C, C, 3, 3, 3) -> [fp16 fragments of w_hi | fp8 fragments of (w_lo 2^(b+11), w 2^b)];
 * *scale_exp = b, the largest power of two with 2^b max|w| <= 240.  Bytes needed / written. */
int64_t sk_conv3d_pack_weight_mix8_host(const float* w_host, int cout, int cin, void* dst_host, int* scale_exp);
/* sk_conv3d_stem_apply storing the mix8 line [hi | x8 | lo8]: out (B, Xt, Yt, Zt, 64 halves). */
int sk_conv3d_stem_apply_mix8(int B, int Xt, int Yt, int Zt, const float* weight, const float* bias,
                              const float* affine, void* out, int cout, const void* workspace,
                              void* stream);
/* GroupNorm affine + SiLU in place, RAW split pair [hi | lo] in -> mix8 line out (C = 32 | 64 | 128). */
int sk_groupnorm_silu_mix8(void* x, const float* affine, int B, int64_t voxels, int C, void* stream);
/* sk_conv3d_stem_apply storing the activation as a split pair: out (B, Xt, Yt, Zt, 64). */
int sk_conv3d_stem_apply_split(int B, int Xt, int Yt, int Zt, const float* weight, const float* bias,
                               const float* affine, void* out, int cout, const void* workspace,
                               void* stream);
/* Fused GroupNorm affine + SiLU in place on a split tensor (B, voxels, 2C), fp32 arithmetic. */
int sk_groupnorm_silu_split(void* x, const float* affine, int B, int64_t voxels, int C, void* stream);
/* sk_heads on a split tensor x (B, X, Y, Z, 2C); out5 stays (B, 5, X, Y, Z) fp16 (the reference's
 * autocast output dtype, eval.py:142-147). */
int sk_heads_split(const void* x, const float* affine, const float* weight, const float* bias,
                   void* out5, int B, int X, int Y, int Z, int C, const int* box_lo_host,
                   const int* box_hi_host, void* stream);

/* fp32 precision mode (parity reference of the fast path): the same layers with fp32 activations
 * (B, x, y, z, C) and torch-layout fp32 weights (cout, cin, k, k, k) on the exact-fp32 matrix
 * instruction.  Same source / upsample / concat semantics and GroupNorm partial layout as
 * sk_conv3d (sources must be activated: affine == NULL); any cout when gn_partial is NULL. */
int sk_conv3d_f32(const sk_conv_src* srcs, int n_src, const float* weight, const float* bias,
                  float* out, int B, int ox, int oy, int oz, int cout, int ksize,
                  float* gn_partial, void* stream);
int sk_conv3d_f32_num_blocks(int ox, int oy, int oz);
int sk_groupnorm_silu_f32(float* x, const float* affine, int B, int64_t voxels, int C, void* stream);

/* ------------------------------------------------------------------------ *
 * Training step (BASELINE.json configs[4]; skoots/train/engine.py:456-499): forward with saved
 * tensors, fused Tversky losses, backward, AdamW.  fp32 on channels-last (B, x, y, z, C).
 * The forward uses sk_conv3d_f32 + sk_groupnorm_finalize_stats + sk_train_gn_silu.
 * ------------------------------------------------------------------------ */

/* sk_groupnorm_finalize that also keeps stats (B, groups, 2) = (mean, rstd) for the backward. */
int sk_groupnorm_finalize_stats(float* gn_partial, int B, int nblocks, int groups, int C,
                                int64_t voxels, const float* gamma, const float* beta, float eps,
                                float* affine, float* stats, void* stream);

/* z = silu(a*y + b), out of place (y, the raw conv output, is needed again by the backward). */
int sk_train_gn_silu(const float* y, const float* affine, float* z, int B, int64_t voxels, int C,
                     void* stream);

/* Backward of GroupNorm(groups) + SiLU: dz -> dy (may alias dz), dgamma (C), dbeta (C).
 * workspace: sk_train_gn_bwd_workspace_floats(B, voxels, C) floats. */
int sk_train_gn_silu_bwd(const float* dz, const float* y, const float* affine, const float* stats,
                         const float* gamma, int B, int64_t voxels, int C, int groups, float* dy,
                         float* dgamma, float* dbeta, float* workspace, void* stream);
int64_t sk_train_gn_bwd_workspace_floats(int B, int64_t voxels, int C);
int sk_train_gn_bwd_num_blocks(int64_t voxels);

/* Fused loss of the step (engine.py:465-493): logits (B, X*Y*Z, 5) = the head conv's raw output
 * (tanh / sigmoid are applied inside); masks, skeleton_masks (B, X*Y*Z) fp32 (> 0 = foreground,
 * engine.py:468-472); baked (B, 3, X*Y*Z) fp32.  Three Tversky terms (train/loss.py:157-209,
 * per sample then batch mean) on: exp(-|E - baked|^2 / 2 sigma^2) with E = index + tanh(l)*scale
 * (lib/embedding_to_prob.py:5-51, lib/vector_to_embedding.py:104-105), the probability map and the
 * skeleton map.  loss_params_host (3, 4) = alpha, beta, eps, relative weight for (embed, prob,
 * skeleton).  losses: DEVICE buffer of 16 floats; [0:4] = embed, prob, skeleton, weighted total.
 * dlogits (B, X*Y*Z, 5) = d total / d logits, or NULL for the forward value only.
 * workspace: sk_train_loss_workspace_floats(B, voxels) floats. */
int sk_train_loss(const float* logits, const float* masks, const float* skeleton_masks,
                  const float* baked, int B, int X, int Y, int Z, const float* vector_scale_host,
                  const float* sigma_host, const float* loss_params_host, float* losses,
                  float* dlogits, float* workspace, void* stream);
int64_t sk_train_loss_workspace_floats(int B, int64_t voxels);
int sk_train_loss_num_blocks(int64_t voxels);

/* baked_embed_to_prob (lib/embedding_to_prob.py:5-51): embedding, baked (B, 3, voxels) fp32 ->
 * out (B, voxels) = exp(sum_k (E_k - S_k)^2 / (-2 (sigma_k + eps)^2)). */
int sk_baked_embed_to_prob(const float* embedding, const float* baked, float* out, int B,
                           int64_t voxels, const float* sigma_host, float eps, void* stream);

/* Stand-alone Tversky value (train/loss.py:95-212): predicted, ground_truth (B, voxels) fp32
 * (ground_truth != 0 = foreground); loss = batch mean of 1 - (TP+eps)/(TP+alpha(FP+1e-10)+beta FN+eps).
 * workspace: sk_train_loss_workspace_floats(B, voxels) floats. */
int sk_train_tversky(const float* predicted, const float* ground_truth, int B, int64_t voxels,
                     float alpha, float beta, float eps, float* loss, float* workspace,
                     void* stream);

/* Data gradient of a conv layer from dy (B, ox, oy, oz, cout) and the layer's own weight
 * (cout, cin_total, k, k, k), read transposed / tap-flipped in place.
 *   ksize 1, 3: dx (B, ox, oy, oz, cin_n) = gradient w.r.t. input channels [cin_lo, cin_lo+cin_n)
 *               (one half of a concatenated input; an upsampled half is pooled afterwards with
 *               sk_train_sumpool2).
 *   ksize 2 (stride 2): dx (B, 2ox, 2oy, 2oz, cin_total); cin_lo = 0, cin_n = cin_total.
 * accumulate != 0 adds into dx (a tensor with two consumers). */
int sk_train_conv_dgrad(const float* dy, const float* weight, float* dx, int B, int ox, int oy,
                        int oz, int cout, int cin_total, int cin_lo, int cin_n, int ksize,
                        int accumulate, void* stream);

/* Weight and bias gradient of a conv layer: srcs as in sk_conv3d_f32 (activated fp32 inputs, the
 * second optionally nearest-upsampled), dy (B, ox, oy, oz, cout) -> dweight (cout, cin, k, k, k),
 * dbias (cout) or NULL.  workspace: sk_train_conv_wgrad_workspace_floats(...) floats. */
int sk_train_conv_wgrad(const sk_conv_src* srcs, int n_src, const float* dy, int B, int ox, int oy,
                        int oz, int cout, int ksize, float* dweight, float* dbias,
                        float* workspace, void* stream);
int64_t sk_train_conv_wgrad_workspace_floats(int B, int ox, int oy, int oz, int cout, int cin,
                                             int ksize);

/* Mixed-precision pieces (TrainUNet precision="mixed"): fp16 copies feed the fp16 MFMA kernels.
 * sk_train_absmax_scale: scale (3 floats, device) <- [2^k, 2^-k, scratch] with max|x| * 2^k in [2^12, 2^13).
 * sk_train_cast_f32_f16: y = fp16(x * scale[0]) (scale NULL: 1).  sk_train_cast_f16_f32: y (+)= float(x) * scale[1].
 * sk_train_conv_wgrad_f16: sk_train_conv_wgrad with fp16 sources and fp16 dy (scaled by dy_scale[0], or
 *   unscaled if dy_scale is NULL) on v_mfma_f32_32x32x16_f16; fp32 partial sums, result multiplied by
 *   dy_scale[1].  Same workspace size as the fp32 entry point.  zero_page (>= 1 KiB of zeros, or NULL):
 *   with it and channel counts % 32 == 0 the operands move as whole 64-byte lines (LDS-DMA + ds_read_b64_tr_b16).
 * sk_train_pack_weight: device-side sk_conv3d_pack_weight_host of the CURRENT fp32 weight (Co, Ci, k, k, k);
 *   transposed != 0 packs the data-gradient operator of input channels [c_lo, c_lo + c_n) instead
 *   (rows = those input channels, K = Co, taps flipped).  dst: k^3 * (cin_eff/16) * (cout_eff/32) KiB.
 */
int sk_train_pack_weight(const float* weight, int Co, int Ci, int ksize, int transposed, int c_lo,
                         int c_n, void* dst, void* stream);
/* transposed == 2: transposed WITHOUT the tap flip -- the stride-2 (ksize 2) data gradient in scatter form:
 * fragment block p (of 8, each (Co/16)*(Ci/32) KiB) is the pointwise operator W_p^T of parity p; run it with
 * sk_conv3d(ksize 1) into t16[p] (B, cx, cy, cz, Ci) and assemble with sk_train_interleave2:
 * dx (B, 2cx, 2cy, 2cz, Ci) (+)= t16[parity][coarse voxel] * scale[1]. */
int sk_train_interleave2(const void* t16, float* dx, int B, int cx, int cy, int cz, int C,
                         const float* scale, int accumulate, void* stream);
/* Lean mixed data flow: sk_train_gn_silu_f16: raw fp16 conv output -> z16 (and z32 if not NULL).
 * sk_train_gn_silu_bwd_f16: dz (fp32) and the RAW fp16 output -> dy16 = fp16(dy * scale[0]) with scale (3 floats,
 * device: 2^k, 2^-k, bound) chosen from an upper bound of max|dy| formed in the reduction pass; dgamma, dbeta as
 * sk_train_gn_silu_bwd.  workspace: sk_train_gn_bwd_f16_workspace_floats(B, voxels, C) floats. */
int sk_train_gn_silu_f16(const void* y16, const float* affine, void* z16, float* z32, int B,
                         int64_t voxels, int C, void* stream);
int sk_train_gn_silu_bwd_f16(const float* dz, const void* y16, const float* affine, const float* stats,
                             const float* gamma, int B, int64_t voxels, int C, int groups, void* dy16,
                             float* scale, float* dgamma, float* dbeta, float* workspace, void* stream);
int64_t sk_train_gn_bwd_f16_workspace_floats(int B, int64_t voxels, int C);
int sk_train_absmax_scale(const float* x, int64_t n, float* scale, void* stream);
int sk_train_cast_f32_f16(const float* x, void* y, int64_t n, const float* scale, void* stream);
int sk_train_cast_f16_f32(const void* x, float* y, int64_t n, const float* scale, int accumulate,
                          void* stream);
int sk_train_conv_wgrad_f16(const sk_conv_src* srcs, int n_src, const void* dy, const float* dy_scale,
                            int B, int ox, int oy, int oz, int cout, int ksize, float* dweight,
                            float* dbias, float* workspace, const void* zero_page, void* stream);

/* coarse (B, cx, cy, cz, C) = 2x2x2 block sums of fine (B, 2cx, 2cy, 2cz, C). */
int sk_train_sumpool2(const float* fine, float* coarse, int B, int cx, int cy, int cz, int C,
                      void* stream);
/* sk_train_gn_silu_bwd_f16h: sk_train_gn_silu_bwd_f16 with the incoming gradient itself a scaled fp16 tensor (the
 *   output of the consumer's fast data-gradient conv): true dz = dz16 * dz_scale[1].  No fp32 copy in between.
 * sk_train_sumpool2_f16: sk_train_sumpool2 from a scaled fp16 fine tensor: coarse = scale[1] * sum of the 8 children. */
int sk_train_gn_silu_bwd_f16h(const void* dz16, const float* dz_scale, const void* y16, const float* affine,
                              const float* stats, const float* gamma, int B, int64_t voxels, int C, int groups,
                              void* dy16, float* scale, float* dgamma, float* dbeta, float* workspace, void* stream);
int sk_train_sumpool2_f16(const void* fine16, const float* scale, float* coarse, int B, int cx, int cy, int cz, int C,
                          void* stream);
/* The stem as a fast block of the mixed step.  sk_train_stem_fwd_f16: image (B, X, Y, Z) fp32 (rounded to fp16 as the
 *   MFMA operand, weights exact through a hi + lo split), weight_t (27, 32) fp32 tap-major, bias (32) -> RAW y16
 *   (B, X, Y, Z, 32) fp16 + gn_partial (B, sk_conv3d_stem_num_blocks(X, Y, Z), 8, 2); workspace:
 *   sk_conv3d_stem_workspace_bytes(B, X, Y, Z).  B <= 32, Z even.
 * sk_train_stem_wgrad_f16: image fp32 and the scaled fp16 dy (B, X, Y, Z, 32) -> dweight (32, 1, 3, 3, 3), dbias (32),
 *   multiplied by dy_scale[1]; workspace: sk_train_conv_wgrad_workspace_floats(B, X, Y, Z, 32, 1, 3). */
int sk_train_stem_fwd_f16(const float* image, int B, int X, int Y, int Z, const float* weight_t, const float* bias,
                          void* y16, float* gn_partial, void* workspace, size_t workspace_bytes, void* stream);
int sk_train_stem_wgrad_f16(const float* image, const void* dy16, const float* dy_scale, int B, int X, int Y, int Z,
                            float* dweight, float* dbias, float* workspace, void* stream);
/* Heads of the mixed step on the fp16 activation z16 (nvox, 32): logits (nvox, 5) fp32 = z W^T + b with W (5, 32);
 * weight / bias gradients from z16 and dlogits (nvox, 5) fp32 (deterministic two-stage reduction; workspace:
 * sk_train_heads_wgrad_workspace_floats(nvox) floats).  nvox counts the voxels of all batch items. */
/* sk_train_interleave2_add16: sk_train_interleave2 (no accumulate) plus a second scaled fp16 gradient of the same fine
 * tensor, add16 (B, 2cx, 2cy, 2cz, C) * add_scale[1] -- the two contributions to a skip tensor in one pass. */
int sk_train_interleave2_add16(const void* t16, const void* add16, const float* add_scale, float* dx, int B, int cx,
                               int cy, int cz, int C, const float* scale, void* stream);
/* 16-bit hand-off of a gradient to the GroupNorm backward that consumes it (sk_train_gn_silu_bwd_f16h): the producer
 * writes the scaled 16-bit tensor AND its scale vector out_scale (3 floats, device) = [s, 1/s, bound], s a power of two
 * derived from the input scales so that the result cannot overflow (no max pass, no fp32 tensor).
 * sk_train_interleave2_h: sk_train_interleave2 / _add16 (add16 may be NULL) with a 16-bit result dx16 (B, 2cx, 2cy, 2cz, C).
 * sk_train_sumpool2_hh: sk_train_sumpool2_f16 with a 16-bit result coarse16 (B, cx, cy, cz, C), s = scale[0] / 8.
 * sk_train_heads_dgrad_f16: data gradient of the heads, dx[v][c] = sum_k dlogits[v][k] W[k][c] (W (5, C) fp32), from the
 *   fp32 dlogits (nvox, 5) and dl_scale = sk_train_absmax_scale(dlogits).  C % 8 == 0. */
int sk_train_interleave2_h(const void* t16, const void* add16, const float* add_scale, void* dx16, float* out_scale, int B,
                           int cx, int cy, int cz, int C, const float* scale, void* stream);
int sk_train_sumpool2_hh(const void* fine16, const float* scale, void* coarse16, float* out_scale, int B, int cx, int cy, int cz,
                         int C, void* stream);
int sk_train_heads_dgrad_f16(const float* dlogits, const float* dl_scale, const float* weight, void* dx16, float* out_scale,
                             int64_t nvox, int C, void* stream);
int sk_train_heads_fwd_f16(const void* z16, const float* weight, const float* bias, float* logits, int64_t nvox,
                           void* stream);
int64_t sk_train_heads_wgrad_workspace_floats(int64_t nvox);
int sk_train_heads_wgrad_f16(const void* z16, const float* dlogits, float* dweight, float* dbias, int64_t nvox,
                             float* workspace, void* stream);

/* One AdamW update (torch.optim.AdamW semantics; engine.py:281-285, config.py:96-101) over a
 * flat parameter buffer; step counts from 1. */
int sk_train_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                   void* stream);

/* ------------------------------------------------------------------------ *
 * Training-target baking (skoots/lib/skeleton.py:448-528 with its Triton kernel :51-251, and
 * average_baked_skeletons :18-48).  masks (X, Y, Z) int32 instance ids; the skeletons as a CSR table: ids (K,
 * ascending), offsets (K+1), points (offsets[K], 3) fp32 voxel coordinates.  baked (3, X, Y, Z) fp32 = the
 * nearest point of the voxel's own skeleton under anisotropy_host (3) (first minimal point on ties, exact
 * distances), 0 for background / unknown ids; distance (X, Y, Z) fp32 or NULL.
 * sk_average_baked_skeletons: out[c, v] = sum of the zero-padded 3x3x3 neighbourhood / number of entries > 0. */
int sk_bake_skeleton(const int32_t* masks, const int32_t* ids, const int32_t* offsets,
                     const float* points, int K, int X, int Y, int Z, const float* anisotropy_host,
                     float* baked, float* distance, void* stream);
int sk_average_baked_skeletons(const float* baked, float* out, int C, int X, int Y, int Z,
                               void* stream);

/* ------------------------------------------------------------------------ *
 * Validation metrics (skoots/validate/lib.py:190-229 mask_iou): iou (N, M) fp32 of the N
 * ground-truth and M predicted instances, intersection / union of voxel counts, 0 for pairs that do not touch.
 * gt, pred (n) int32; lut_gt (max_gt + 1) / lut_pred (max_pred + 1) int32 map an id to its 1-based row / column
 * (0 = ignore: background or unlisted).  workspace: sk_mask_iou_workspace_bytes(N, M). */
size_t sk_mask_iou_workspace_bytes(int N, int M);
int sk_mask_iou(const int32_t* gt, const int32_t* pred, int64_t n, const int32_t* lut_gt, int max_gt, int N,
                const int32_t* lut_pred, int max_pred, int M, float* iou, void* workspace,
                size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------ *
 * Diagnostics (no reference counterpart; not on the hot path)
 * ------------------------------------------------------------------------ */

/* Box probe: one launch of a bare v_mfma_f32_16x16x32_f16 register loop (512 workgroups of 4 waves, `iters` x 8
 * MFMAs per wave, pseudo-random operands).  vary_operands = 0: every MFMA multiplies the same register pair;
 * 1: four A x four B fragments in rotation, so that the operand data changes with every instruction as in a real
 * kernel (the chip holds a lower clock then: the rate a matrix kernel on real data can reach).  The caller times it
 * with events on `stream`; *flops (host, may be NULL) receives the FLOPs of the launch.  bench.py prints both rates
 * next to its line: devices differ, so figures from two boxes compare only beside them.  scratch: >= 512 KiB. */
int sk_mfma_probe(void* scratch, size_t scratch_bytes, int iters, int vary_operands, double* flops, void* stream);

/* CU-masked streams (round 4 experiment, tools/cu_*_probe.py; no reference counterpart: the reference runs everything on
 * the default stream of one device, eval.py:57; NOT used by the product path -- DESIGN.md section 8 records why).
 * sk_stream_create_cu_mask wraps hipExtStreamCreateWithCUMask (the stream must belong to the HIP runtime instance the
 * kernels are launched from): bit i of the little-endian word array enables a compute unit; on MI355X bit 8 c + x is
 * compute unit c of XCD x, and an XCD with no bit set gets all its CUs.  sk_debug_where launches n_blocks one-wave
 * workgroups that each record (XCC id, HW id register) after spinning spin_cycles: which CUs a mask really selects. */
int sk_stream_create_cu_mask(const uint32_t* mask_words, int n_words, void** stream_out);
int sk_stream_destroy(void* stream);
int sk_debug_where(unsigned* out, int n_blocks, int spin_cycles, void* stream);

/* Phase-timing builds (-DSK_TIMING, tools/conv_phase_timing.py) dump per-wave cycle sums of the conv kernels into
 * this device buffer, [4096 workgroups][4 waves][16 slots] int64 (bytes must cover all of it); NULL detaches it.
 * The release library stores the pointer and never writes through it. */
int sk_debug_set_timing_buffer(void* device_ptr, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* SKOOTS_HIP_H */
