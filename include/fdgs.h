/*
 * fdgs.h -- C-ABI of the MI355X-native 4D-Gaussian render path (libfdgs.so, built by hipcc for gfx950).
 *
 * Drop-in boundary.  The reference binds its rasterizer through the pybind module
 * `diff_gaussian_rasterization._C` of the un-vendored submodule named at .gitmodules:5-7 of the reference
 * (call sites gaussian_renderer/__init__.py:14,38-58,120-128; merge_many_4dgs.py:33,85-135;
 * scene/dataset_readers.py:485-508), and its deformation step through the PyTorch modules
 * scene/deformation.py:161-216 + scene/hexplane.py:109-183.  The entry points below are what a binding of this
 * path needs instead: plain pointers and sizes, an explicit hipStream_t (passed as void*), every buffer owned and
 * allocated by the caller, no torch types, no exceptions across the ABI.
 *
 * Conventions
 *   - every function returns 0 on success, a negative FDGS_E_* code otherwise; fdgs_last_error() returns a
 *     thread-local message for the last failure on the calling thread;
 *   - all device pointers are float32 unless stated; "opt" pointers may be NULL;
 *   - matrices are the reference's transposed 4x4s (scene/cameras.py:59-63): flat index m[4*col+row];
 *   - no function synchronises the device except fdgs_bin_prepare (the single num_rendered read-back that the
 *     reference's rasterizer also performs) and the *_debug_sync helpers;
 *   - functions are re-entrant per stream; one host thread per stream.
 */
#ifndef FDGS_H
#define FDGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDGS_OK 0
#define FDGS_E_INVALID (-1) /* bad argument / unsupported configuration */
#define FDGS_E_HIP (-2)     /* a HIP runtime call or kernel launch failed */
#define FDGS_E_NOGPU (-3)   /* no gfx950 device visible */

#define FDGS_TILE 16         /* tile edge in pixels (BLOCK_X = BLOCK_Y of the reference rasterizer) */
#define FDGS_SH_COEFFS 16    /* max SH coefficients per channel (degree 3) */

const char* fdgs_last_error(void);
/* ABI version of this header; bump on any signature change. */
int fdgs_abi_version(void);
/* Optional per-kernel timing with HIP events on the launch stream (off by default; used by bench.py for the live
 * roofline measurement).  fdgs_timing_report synchronises the device and writes "kernel_name count total_ms" lines. */
int fdgs_timing_enable(int on);
int fdgs_timing_report(char* buf, size_t buflen, int reset);
/* Development / test knobs (ABI 4; ten since ABI 5, eleven since round 6).  The library holds ONE table of integer knobs; it is filled from the environment variables
 * FDGS_<NAME> once, when the library is loaded, and afterwards changes only through fdgs_tuning_set -- no entry point reads the
 * environment.  Knobs select between equivalent kernel forms or launch shapes (same results up to summation order), never semantics:
 *   d1_form (16 | 32), d1_wgs, d1_split, skip_dead, d4_mfma, d4_rows_kb, tile_cull (0 = the reference's rectangle lists), rbwd_ppl (-1 = by image size | 4 | 2 | 0),
 *   tile_order (1 = the blending kernels take their tiles heaviest-first, 0 = image order), row_compact (1 = the deformation backward walks
 *   the non-zero rows instead of the non-zero 32-row tiles where it can; 2 = the same with the row lists built by the two-launch form),
 *   d2_form (0 = the weight-stationary backward-data kernel where it applies -- row lists, net_width 128, C*L 32 or 48, five heads or position + scale + rotation --, 32 = the
 *   32-row kernel everywhere).
 * `name` is the lower- or upper-case knob name, with or without the FDGS_ prefix.  Process-global, not thread-safe against running calls:
 * set knobs between frames.  fdgs_tuning_reset restores the load-time values.  See INTEGRATION.md ("Knobs"). */
int fdgs_tuning_set(const char* name, int value);
int fdgs_tuning_get(const char* name, int* value);
int fdgs_tuning_reset(void);
/* Name (gcnArchName) of device `dev` into buf; FDGS_E_NOGPU if none. */
int fdgs_device_arch(int dev, char* buf, size_t buflen);

/* ------------------------------------------------------------------------------------------------------------
 * Rasterizer: replaces _C.rasterize_gaussians / _C.rasterize_gaussians_backward / _C.mark_visible
 * ---------------------------------------------------------------------------------------------------------- */

/* Scratch sizes in bytes. geom: per-Gaussian state (kept for backward); img: per-pixel + per-tile state (kept for
 * backward); binning: sorted (tile, Gaussian) lists for `num_rendered` pairs (kept for backward). */
int fdgs_geom_bytes(int P, size_t* bytes);
int fdgs_img_bytes(int W, int H, size_t* bytes);
int fdgs_binning_bytes(uint32_t num_rendered, int W, int H, size_t* bytes);

typedef struct fdgs_raster_params {
    int P;                  /* number of Gaussians */
    int sh_degree;          /* active degree 0..3 */
    int sh_coeffs;          /* coefficients per channel actually stored per Gaussian (M, <=16) */
    int W, H;               /* image size in pixels */
    float tanfovx, tanfovy;
    float scale_modifier;
    int prefiltered;        /* accepted for API parity; ignored like the reference's non-debug path */
    int debug;              /* nonzero: synchronise + check errors after every kernel */
    const float* bg;        /* [3] device */
    const float* viewmatrix;/* [16] device */
    const float* projmatrix;/* [16] device */
    const float* campos;    /* [3] device */
    const float* means3D;   /* [P,3] */
    const float* shs;       /* opt [P,M,3] (exactly one of shs / colors_precomp) */
    const float* colors_precomp; /* opt [P,3] */
    const float* opacities; /* [P] */
    const float* scales;    /* opt [P,3] (scales+rotations xor cov3D_precomp) */
    const float* rotations; /* opt [P,4] (w,x,y,z), used as given (no normalisation) */
    const float* cov3D_precomp; /* opt [P,6] */
    uint8_t* visibility;    /* opt [P], written by fdgs_preprocess_fwd: 1 where radii > 0 (the `visibility_filter` render() returns,
                               gaussian_renderer/__init__.py:136) -- saves the caller one elementwise launch per frame */
    float* acc_zero;        /* opt [P,16] floats (ABI 5): fdgs_render_fwd zero-fills them on the way.  A training frame passes the buffer it will
                               hand to fdgs_raster_bwd as fdgs_raster_grads::scratch_acc, with scratch_acc_zeroed = 1: the 64 bytes per Gaussian
                               the blending backward accumulates into are then clean without a fill launch between the loss and the backward */
} fdgs_raster_params;

/* Stage 1: per-Gaussian projection (frustum cull, cov3D, EWA cov2D, conic, radius, tile rect, SH->RGB).
 * Writes radii[P] (int32) and fills `geom`. */
int fdgs_preprocess_fwd(void* stream, const fdgs_raster_params* p, void* geom, int32_t* radii);

/* Stage 2: depth-sorts the Gaussians (LSD radix on the fp32 depth bits), scans tiles_touched in depth order and
 * reads the total back to the host: *num_rendered_host = number of (tile, Gaussian) pairs.  This is the one
 * blocking read-back of the path (the reference's rasterizer has the same one). */
int fdgs_bin_prepare(void* stream, const fdgs_raster_params* p, void* geom, uint32_t* num_rendered_host);

/* Stage 3: writes the pairs in depth order, stable-sorts them by tile id (LSD radix), finds per-tile ranges. */
int fdgs_bin_sort(void* stream, const fdgs_raster_params* p, void* geom, void* binning, void* img,
                  uint32_t num_rendered);

/* Stage 4: front-to-back alpha blending per 16x16 tile. out_color [3,H,W], out_depth [1,H,W]. */
/* Capacity mode (ABI 4): stages 1-4 in ONE call and WITHOUT a host synchronisation.  `binning` holds fdgs_binning_bytes(capacity)
 * bytes for a pair count the caller predicts (e.g. 1.5 x the largest count it has seen); the true count stays on the device -- the kernels
 * work on min(true, capacity) pairs -- and reaches *num_rendered_host (pinned host memory) asynchronously (ABI 6: a store of the LAST
 * workgroup of the projection kernel when the word is device-visible -- hipHostGetDevicePointer --, i.e. while the depth sort is still
 * running; a copy node otherwise): pre-set the word to 0xFFFFFFFF and read it once it changed (fdgs_pair_count_wait).  true > capacity
 * means the FARTHEST pairs of this frame were dropped (pairs are emitted in depth order) and the image is NOT the reference's: the
 * stage-1/2 results in `geom` are complete and valid, so the caller finishes the frame exactly with fdgs_bin_sort + fdgs_render_fwd on a
 * buffer of fdgs_binning_bytes(true) bytes (what the Python host does before it hands the image to anyone: the reference never truncates,
 * SURVEY Appendix B.2).  The buffers are laid out for `capacity`: pass `capacity` as num_rendered to fdgs_raster_bwd /
 * fdgs_binning_field.  P = 0 writes 0 to the word immediately. */
int fdgs_raster_fwd_capacity(void* stream, const fdgs_raster_params* p, void* geom, void* binning, void* img, uint32_t capacity,
                             uint32_t* num_rendered_host, int32_t* radii, float* out_color, float* out_depth);
/* ABI 6: waits (spinning on the pinned word, no hipStreamSynchronize) until the pair count of a frame queued by fdgs_raster_fwd_capacity on
 * `stream` has arrived, and returns it in *value.  The rest of the frame stays queued behind it, so the device does not drain while the
 * host looks at the count.  Error if the stream runs idle without a count (wrong stream / word). */
int fdgs_pair_count_wait(void* stream, const uint32_t* num_rendered_host, uint32_t* value);
int fdgs_render_fwd(void* stream, const fdgs_raster_params* p, const void* geom, const void* binning, void* img,
                    uint32_t num_rendered, float* out_color, float* out_depth);

/* Optional epilogue of fdgs_raster_bwd for the fused render() path (deformation -> rasterizer): the per-Gaussian chain rule
 * writes its results straight in the form fdgs_deform_bwd consumes -- the packed pre-activation output-gradient rows G[Npad][64]
 * (columns 0-2 position, 3-5 log-scale, 6-9 raw quaternion, 10 opacity logit, 16-63 SH; activation Jacobians of
 * gaussian_renderer/__init__.py:97-99 applied when `activate`) and the identity paths of `out = in + delta` accumulated into the
 * parameter gradients -- instead of dL_dmeans3D / dL_dscales / dL_drotations / dL_dopacity / dL_dsh, which are then not written.
 * G is the START of the scratch buffer later handed to fdgs_deform_bwd (fdgs_deform_grads::scratch) with packed_rows_ready = 1 / 2 / 3
 * for tile_flags = 0 / 1 / 2.
 * Saves one kernel and a write + read of 236 bytes per Gaussian between the two backward stages. */
typedef struct fdgs_raster_deform_epilogue {
    int activate;                /* the scales / rotations / opacities the rasterizer received are exp / normalize / sigmoid outputs */
    int Npad;                    /* rows of G: fdgs_deform_bwd's padded Gaussian count (multiple of 128, >= P) */
    const float* rot_norm;       /* [P] norm of the raw quaternion (activate = 1), fdgs_deform_out::rot_norm */
    float* G;                    /* [Npad,64], every row written (rows >= P zero) unless tile_flags = 2 */
    float* d_xyz; float* d_scales; float* d_rotations; float* d_opacity;   /* identity paths (any may be NULL) */
    float* d_shs_dc; float* d_shs_rest;                                    /* strides in floats as in fdgs_deform_params */
    int shs_dc_stride; int shs_rest_stride;
    int assign;                  /* 0: the identity paths are accumulated (+=, caller zero-fills); 1: they are ASSIGNED (=): the caller
                                    needs no zero fill of these six arrays -- fdgs_deform_bwd, which runs afterwards, only ever adds to
                                    d_xyz (the HexPlane coordinate gradient) */
    int tile_flags;              /* 1: G is followed by uint32 tile_live[Npad/32] (i.e. at (uint32_t*)(G + Npad*64)), written here:
                                    bit r of tile_live[t] is set when packed row 32t + r has a non-zero entry (a tile is live when its word is
                                    non-zero; ABI 5: the word is the row mask, it used to be 0 / 1).  Culled, occluded and
                                    off-screen Gaussians get all-zero rows (the reference gives them no gradient either,
                                    gaussian_renderer/__init__.py:134-138); fdgs_deform_bwd (packed_rows_ready = 2) then skips whole tiles of
                                    them -- exact, a zero row adds exactly zero to every sum -- and, with saved activations on spatially ordered
                                    input, walks only the non-zero ROWS (tuning knob row_compact).
                                    2: as 1, and the 32 rows of a tile flagged 0 are NOT written (their content is unspecified): the caller
                                    MUST hand G to fdgs_deform_bwd with packed_rows_ready = 3, which never reads them (the one dead tile it
                                    may use as padding is zero-filled there).
                                    0: no flags (packed_rows_ready = 1: fdgs_deform_bwd walks every tile) */
    float* zero_fill;            /* opt: a float range this kernel zero-fills on the way (16-byte aligned; zero_floats a multiple of 4): the part
                                    of the caller's gradient arena that fdgs_deform_bwd ACCUMULATES into (planes, MLP) -- saves a fill launch */
    size_t zero_floats;
} fdgs_raster_deform_epilogue;

typedef struct fdgs_raster_grads {
    const float* dL_dcolor;  /* [3,H,W] */
    const float* dL_ddepth;  /* opt [1,H,W] */
    /* outputs: every row is written by fdgs_raster_bwd (zeros for culled Gaussians); no caller-side zero fill needed */
    float* dL_dmeans2D;      /* [P,3]: x,y in NDC units, z = 0 (the reference's viewspace_points.grad) */
    float* dL_dmeans3D;      /* [P,3] */
    float* dL_dopacity;      /* [P] */
    float* dL_dcolors;       /* [P,3] gradient of the blended per-Gaussian rgb (== dL/dcolors_precomp) */
    float* dL_dsh;           /* opt [P,M,3] (required when shs was given) */
    float* dL_dscales;       /* opt [P,3] */
    float* dL_drotations;    /* opt [P,4] */
    float* dL_dcov3D;        /* [P,6] */
    /* scratch owned by the caller: [P,16] floats (one 64-byte gradient line per Gaussian, see render.hip) */
    float* scratch_acc;
    /* opt: see above (requires shs + scales/rotations inputs, 16 SH coefficients) */
    const fdgs_raster_deform_epilogue* deform_epilogue;
    int scratch_acc_zeroed;  /* 1: scratch_acc is the buffer the forward of THIS state zero-filled (fdgs_raster_params::acc_zero) and no backward
                                has used it since; 0: fdgs_raster_bwd fills it */
} fdgs_raster_grads;

/* Backward of stages 4 and 1 (back-to-front blending gradients, then per-Gaussian chain rule). */
int fdgs_raster_bwd(void* stream, const fdgs_raster_params* p, const void* geom, const void* binning,
                    const void* img, uint32_t num_rendered, const fdgs_raster_grads* g);

/* Frustum test only (replaces _C.mark_visible): present[P] (uint8) = 1 where p_view.z > 0.2. */
int fdgs_mark_visible(void* stream, int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                      uint8_t* present);

/* Test/diagnostic access to geom fields (device pointers into `geom`):
 * 0 depth f32[P]; 1 recA float4[P] (x,y,conic.xx,conic.xy); 2 recB float4[P] (conic.yy,opacity,depth,0);
 * 3 recC float4[P] (r,g,b,0); 4 cov3D f32[P,6]; 5 tiles_touched u32[P]; 6 clamped u32[P] (bit c = channel c);
 * 7 rect u32[P,2] (xmin|ymin<<16, xmax|ymax<<16); 8 sorted Gaussian ids u32[P]; 9 point_offsets u32[P] (inclusive, depth
 * order; ABI 5: LOCAL to chunks of 4096 Gaussians -- the pair expansion adds the totals of the chunks in front itself). */
int fdgs_geom_field(void* geom, int P, int which, void** ptr);
/* 0 sorted pair Gaussian ids u32[R]; 1 sorted pair tile ids u32[R]. */
int fdgs_binning_field(void* binning, uint32_t num_rendered, int W, int H, int which, void** ptr);
/* 0 final_T f32[H*W]; 1 n_contrib u32[H*W]; 2 ranges u32[ntiles,2]; 3 per-tile walk length of the backward u32[ntiles] (max n_contrib);
 * 4 / 5 heaviest-first tile order of the forward / backward, u32[8][per_xcd] with per_xcd = ceil(ceil(gy / 2) / 8) * 2 * gx (0xFFFFFFFF =
 * padding slot; written only while the tuning knob tile_order is on and per_xcd <= 8192). */
int fdgs_img_field(void* img, int W, int H, int which, void** ptr);

/* ------------------------------------------------------------------------------------------------------------
 * Deformation field: replaces deform_network.forward (scene/deformation.py:185-212) + the activations of
 * gaussian_renderer/__init__.py:97-99 when `activate` is set.
 * ---------------------------------------------------------------------------------------------------------- */

#define FDGS_MAX_LEVELS 4
#define FDGS_HEAD_POS 0
#define FDGS_HEAD_SCALE 1
#define FDGS_HEAD_ROT 2
#define FDGS_HEAD_OPACITY 3
#define FDGS_HEAD_SHS 4
#define FDGS_NUM_HEADS 5

typedef struct fdgs_deform_params {
    int N;                 /* Gaussians */
    int C;                 /* features per plane (output_coordinate_dim: 16 or 32) */
    int L;                 /* levels (len(multires) <= 4) */
    int W;                 /* net_width (64 or 128) */
    int head_on[FDGS_NUM_HEADS]; /* 1 = head active (not no_dx / no_ds / no_dr / no_do / no_dshs) */
    int activate;          /* 1: outputs are exp(scale), normalize(rot), sigmoid(opacity) */
    int res[FDGS_MAX_LEVELS][4]; /* per level: resolution of axes x,y,z,t */
    /* planes: channel-LAST device arrays [res_j][res_i][C] for pair k=(i,j) in order (0,1),(0,2),(0,3),(1,2),(1,3),(2,3) */
    const float* planes[FDGS_MAX_LEVELS][6];
    float aabb[6];         /* aabb[0..2] = "max" row, aabb[3..5] = "min" row of the reference's HexPlaneField.aabb */
    const float* w0; const float* b0;               /* feature_out.0: [W, C*L], [W] */
    const float* w1[FDGS_NUM_HEADS]; const float* b1[FDGS_NUM_HEADS]; /* <head>.1: [W,W],[W] */
    const float* w2[FDGS_NUM_HEADS]; const float* b2[FDGS_NUM_HEADS]; /* <head>.3: [k,W],[k], k = 3,3,4,1,48 */
    /* per-Gaussian inputs */
    const float* xyz;      /* [N,3] */
    const float* scales;   /* [N,3] log-scales */
    const float* rotations;/* [N,4] */
    const float* opacity;  /* [N,1] logits */
    const float* shs_dc;   /* features_dc:   3 floats per Gaussian at shs_dc[n*shs_dc_stride + m] */
    const float* shs_rest; /* features_rest: 45 floats per Gaussian at shs_rest[n*shs_rest_stride + m] */
    int shs_dc_stride;     /* 3 for a separate [N,1,3] tensor; 48 when both point into one [N,16,3] tensor */
    int shs_rest_stride;   /* 45 for a separate [N,15,3] tensor; 48 (with shs_rest = shs + 3) for a combined one */
    const float* time;     /* [N] per-Gaussian time, or NULL to use time_scalar for every Gaussian */
    float time_scalar;
} fdgs_deform_params;

typedef struct fdgs_deform_out {
    float* xyz;       /* [N,3] */
    float* scales;    /* [N,3] */
    float* rotations; /* [N,4] */
    float* opacity;   /* [N,1] */
    float* shs;       /* [N,16,3] */
    float* rot_norm;  /* opt [N]: |r + dr| before normalisation (written when activate=1; the backward needs it) */
    void* saved;      /* opt, fdgs_deform_saved_bytes() bytes: when given, the forward also stores the HexPlane features,
                         relu(hidden) and relu(h1) of every active head; handing the same buffer to fdgs_deform_bwd lets the
                         backward skip the gather and the recomputation of those layers (about 40 % of its MFMA work) */
    void* packed;     /* opt, fdgs_deform_pack_bytes() bytes of scratch: when given (and C % 16 == 0), the forward first re-orders W0 and the
                         heads' W1 into it as the operand streams of its matrix-core loops (contiguous 1-KB loads instead of 64 cache lines
                         per request) and runs its 16-Gaussians-per-wave form, two waves per SIMD -- the faster form inside a frame
                          and what the Python host hands over by default; NULL: the 32-Gaussian form on the row-major
                         weights.  Same results up to summation order, same `saved` format.  (Tuning knob d1_form = 32 forces the
                         32-Gaussian form for A/B runs.) */
} fdgs_deform_out;

int fdgs_deform_saved_bytes(const fdgs_deform_params* p, size_t* bytes);
int fdgs_deform_pack_bytes(const fdgs_deform_params* p, size_t* bytes);
int fdgs_deform_fwd(void* stream, const fdgs_deform_params* p, const fdgs_deform_out* out);

typedef struct fdgs_deform_grads {
    /* incoming gradients w.r.t. the outputs of fdgs_deform_fwd (same `activate` setting); any may be NULL (=0) */
    const float* g_xyz; const float* g_scales; const float* g_rotations; const float* g_opacity; const float* g_shs;
    /* forward outputs (needed when activate=1 for the activation Jacobians) */
    const float* out_scales; const float* out_rotations; const float* out_opacity; const float* rot_norm;
    /* outgoing gradients; ACCUMULATED into (+=): the caller zero-fills.  Any may be NULL (skipped). */
    float* d_xyz; float* d_scales; float* d_rotations; float* d_opacity;
    float* d_shs_dc; float* d_shs_rest; /* laid out with the strides of the inputs (shs_dc_stride / shs_rest_stride) */
    float* d_planes[FDGS_MAX_LEVELS][6]; /* channel-last like `planes` */
    float* d_w0; float* d_b0;
    float* d_w1[FDGS_NUM_HEADS]; float* d_b1[FDGS_NUM_HEADS];
    float* d_w2[FDGS_NUM_HEADS]; float* d_b2[FDGS_NUM_HEADS];
    /* scratch owned by the caller, fdgs_deform_bwd_scratch_bytes() bytes */
    void* scratch;
    /* opt: the `saved` buffer the forward of the SAME parameters / inputs filled (NULL: everything is recomputed) */
    const void* saved;
    /* 0: fdgs_deform_bwd packs the rows (and computes the per-tile flags) itself.
     * 1: `scratch` already starts with the packed gradient rows and the identity paths were applied (fdgs_raster_bwd's epilogue,
     *    tile_flags = 0); the g_* / out_* / rot_norm pointers above are then ignored.  No flags: every tile is walked.
     * 2: as 1, and the rows are followed by the per-tile non-zero flags (epilogue tile_flags = 1): all-zero tiles are skipped.
     * 3: as 2, and the rows of tiles flagged 0 were NOT written (epilogue tile_flags = 2): all-zero tiles are ALWAYS skipped.
     * (The tuning knob skip_dead = 0 makes modes 0 and 2 walk every tile for A/B runs; it cannot affect modes 1 and 3.) */
    int packed_rows_ready;
    /* hint, never needed for correctness: 1 = consecutive Gaussians are spatial neighbours (the set is kept along a space-filling
     * curve, fdgs.densify.spatial_reorder): with one frame time for all Gaussians the plane gradient then runs as the windowed
     * matrix-core splat (one atomic line per touched texel of a chunk's window); 0 = unknown order: per-corner atomics, which are
     * faster on unordered input (0.41 vs 0.72 ms at 300 k) and slower on ordered input (0.69 vs 0.31 ms) */
    int spatially_ordered;
} fdgs_deform_grads;

int fdgs_deform_bwd_scratch_bytes(const fdgs_deform_params* p, size_t* bytes);
int fdgs_deform_bwd(void* stream, const fdgs_deform_params* p, const fdgs_deform_grads* g);
/* Diagnostics (bench.py's FLOP accounting): after fdgs_deform_bwd on `scratch`, out_host[0] = 32-row tiles the backward processed
 * (tiles with a non-zero packed gradient row, + at most 3 of padding; in the row-list form -- tuning knob row_compact, saved activations,
 * spatially ordered input -- the 32-row units of the list of non-zero ROWS, padded to whole chunks), out_host[1] = tiles in total (Npad / 32),
 * out_host[2] = plane-gradient chunks processed, out_host[3] = chunks in total.  Synchronises the stream. */
int fdgs_deform_bwd_live_tiles(void* stream, const fdgs_deform_params* p, const void* scratch, uint32_t* out_host);

/* ------------------------------------------------------------------------------------------------------------
 * Loss-side helper for the frame-parallel driver: accumulates [sum|a-b|, sum (a-b)^2, n] into acc[3] (device,
 * caller zero-fills) and optionally writes dL/da = sign(a-b)*scale.  (utils/loss_utils.py:20-21, image_utils.py:17-38)
 * ---------------------------------------------------------------------------------------------------------- */
int fdgs_l1_stats(void* stream, size_t n, const float* a, const float* b, float grad_scale, float* grad_out_opt,
                  float* acc);
/* Same statistics ASSIGNED to acc[3] (no zero fill by the caller, no same-address float atomics, sums added in a fixed order: deterministic).
 * `scratch`: fdgs_l1_stats_scratch_bytes() bytes of device memory, zero-filled ONCE at allocation (it holds a self-resetting ticket counter);
 * one scratch per stream. */
int fdgs_l1_stats_scratch_bytes(size_t* bytes);
int fdgs_l1_stats_assign(void* stream, size_t n, const float* a, const float* b, float grad_scale, float* grad_out_opt,
                         float* acc, void* scratch);

/* ------------------------------------------------------------------------------------------------------------
 * Image losses of the step right after render() (train.py:201-214): l1_loss (utils/loss_utils.py:20-21), the sum of
 * squares psnr() needs (utils/image_utils.py:17-38) and ssim() (utils/loss_utils.py:40-66: 11x11 Gaussian window,
 * sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2), forward value and gradient wrt the rendered image.
 * img / gt: [items][channels][H][W] contiguous f32 (device).
 *   fwd: acc[item][4] += { sum|img-gt|, sum (img-gt)^2, channels*H*W, sum of the SSIM map }   (device, caller zero-fills)
 *        ssim_maps_opt: [3][items*channels][H][W] partial derivatives kept for the backward pass (NULL: value only)
 *   bwd: dimg = s * ( w_l1 * sign(img-gt) + w_ssim * d(sum SSIM map)/d img ),  s = *grad_scale_dev_opt (device) or 1
 *        (loss = L1 + lambda (1 - ssim)  ->  w_l1 = 1/n, w_ssim = -lambda/n with n = items*channels*H*W)
 * ---------------------------------------------------------------------------------------------------------- */
int fdgs_image_loss_fwd(void* stream, int items, int channels, int H, int W, const float* img, const float* gt,
                        float* ssim_maps_opt, float* acc);
int fdgs_image_loss_bwd(void* stream, int items, int channels, int H, int W, const float* img, const float* gt,
                        const float* ssim_maps, float w_l1, float w_ssim, const float* grad_scale_dev_opt, float* dimg);

/* ------------------------------------------------------------------------------------------------------------
 * HexPlane regulariser: replaces GaussianModel.compute_regulation (scene/gaussian_model.py:538-577 with
 * compute_plane_smoothness of scene/regulation.py:22-28; evaluated at train.py:208-211 every fine iteration).
 *   loss += sum_planes [ w_smooth * mean_{C,H-2,W} (p[h+2] - 2 p[h+1] + p[h])^2 + w_l1 * mean |1 - p| ]
 * The caller passes, per plane, the weight that the reference applies to it: planes (0,1,3) of a level get
 * w_smooth = plane_tv_weight, planes (2,4,5) get w_smooth = time_smoothness_weight and w_l1 = l1_time_planes_weight.
 * Planes are channels-last [H][W][C] device arrays (as in fdgs_deform_params).  One launch computes the value
 * (accumulated into *loss_acc_opt, caller zero-fills) and, for planes with grad_opt != NULL, accumulates
 * grad_scale * (*grad_scale_dev_opt, 1 if NULL) * dloss/dplane into grad_opt (same layout).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct fdgs_reg_plane {
    const float* plane;   /* [H][W][C] */
    float* grad_opt;      /* [H][W][C], accumulated (+=) */
    int H, W, C;
    float w_smooth;       /* weight of the second-difference term (0: skipped) */
    float w_l1;           /* weight of mean |1 - p| (0: skipped) */
} fdgs_reg_plane;
int fdgs_plane_regulation(void* stream, int nplanes, const fdgs_reg_plane* planes /* host array */, float grad_scale,
                          const float* grad_scale_dev_opt, float* loss_acc_opt);

/* ------------------------------------------------------------------------------------------------------------
 * Optimizer step: replaces torch.optim.Adam(l, lr=0.0, eps=1e-15).step() of the reference (scene/gaussian_model.py:184,
 * train.py:291) for all parameter tensors of a step in one launch.  Per tensor: element count, the four device
 * arrays (same memory order, 16-byte aligned), the group's learning rate and the tensor's step count AFTER this step
 * (t >= 1, bias corrections 1 - beta^t).  No weight decay, no amsgrad (the reference uses neither).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct fdgs_adam_tensor {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    size_t n;
    float lr;
    int step;
} fdgs_adam_tensor;
int fdgs_adam_step(void* stream, int ntensors, const fdgs_adam_tensor* tensors /* host array */, double beta1, double beta2,
                   double eps);   /* doubles: 1 - beta and the bias corrections are formed in double like torch does */

/* ------------------------------------------------------------------------------------------------------------
 * Scale initialisation: replaces simple_knn._C.distCUDA2 (un-vendored submodule submodules/simple-knn, .gitmodules:1-3;
 * call site scene/gaussian_model.py:148).  mean_dist2[i] = mean of the three smallest squared distances from point i
 * to the other points (self excluded by index).  points [N,3], mean_dist2 [N], device.
 * ---------------------------------------------------------------------------------------------------------- */
int fdgs_knn3_mean_dist2(void* stream, int N, const float* points, float* mean_dist2);

/* ------------------------------------------------------------------------------------------------------------
 * Densification bookkeeping (the consumer of render()'s radii / viewspace_points.grad).
 *
 * fdgs_densification_stats: train.py:259-262 + GaussianModel.add_densification_stats (scene/gaussian_model.py:516-518),
 *   for every visible Gaussian (visibility_opt != 0, or radii > 0 when it is NULL):
 *     max_radii2D = max(max_radii2D, radii)   (skipped when max_radii2D_opt is NULL)
 *     xyz_gradient_accum += |viewspace_grad[:, :2]|,  denom += 1            (viewspace_grad rows of grad_stride floats)
 *
 * fdgs_densify_plan / fdgs_densify_apply: GaussianModel.densify (densify_and_clone + densify_and_split,
 *   scene/gaussian_model.py:409-456, 495-500), GaussianModel.prune / prune_points (:350-365, 481-494) with the
 *   optimizer-state surgery of cat_tensors_to_optimizer / _prune_optimizer (:331-348, 367-389).
 *   plan  FDGS_PLAN_DENSIFY: g = accum/denom (NaN -> 0), big = max(exp(scaling)) > dense_size (= percent_dense * extent):
 *                            clone when |g| >= grad_threshold and not big; split (N = 2) when g >= grad_threshold and big
 *         FDGS_PLAN_PRUNE  : drop when sigmoid(opacity) < min_opacity, or (max_screen_size > 0 and (max_radii2D >
 *                            max_screen_size or max(exp(scaling)) > max_world_size (= 0.1 * extent)))
 *         FDGS_PLAN_MASK   : drop where drop_mask != 0 (prune_points(mask))
 *         counts_host[3] = { rows kept, clones, split Gaussians }: the ONE blocking readback (the reference syncs on
 *         .any() / .sum() / boolean indexing many times); N_out = kept + clones + 2 * splits.
 *   apply writes every output row exactly once, order [kept originals | clones | first children | second children];
 *         exp_avg / exp_avg_sq of new rows are 0 (NULL inputs = no optimizer state yet = zeros; NULL outputs = skipped);
 *         side arrays: deformation_table follows its Gaussian, the four statistics survive on kept rows and are 0 on
 *         new rows (any of them may be NULL in `out`).  split_samples: [2*splits][3] standard normals, row
 *         k*splits + r belongs to child k of the r-th split Gaussian (torch.normal over stds.repeat(2,1)); NULL puts
 *         the children on their parent.
 * ---------------------------------------------------------------------------------------------------------- */
#define FDGS_NGROUPS 6          /* xyz, f_dc, f_rest, opacity, scaling, rotation (scene/gaussian_model.py:176-185) */
#define FDGS_PLAN_DENSIFY 0
#define FDGS_PLAN_PRUNE 1
#define FDGS_PLAN_MASK 2
typedef struct {
    int N;                                  /* Gaussians before */
    int width[FDGS_NGROUPS];                /* floats per row: 3, 3, 3*((deg+1)^2-1), 1, 3, 4 (0 = group absent) */
    const float* param[FDGS_NGROUPS];
    const float* exp_avg[FDGS_NGROUPS];
    const float* exp_avg_sq[FDGS_NGROUPS];
    const uint8_t* deformation_table;       /* opt [N] */
    const float* xyz_gradient_accum;        /* opt [N] */
    const float* denom;                     /* opt [N] */
    const float* max_radii2D;               /* opt [N] */
    const float* deformation_accum;         /* opt [N,3] */
} fdgs_gaussians_in;
typedef struct {
    float* param[FDGS_NGROUPS];             /* [N_out][width] */
    float* exp_avg[FDGS_NGROUPS];
    float* exp_avg_sq[FDGS_NGROUPS];
    uint8_t* deformation_table;
    float* xyz_gradient_accum;
    float* denom;
    float* max_radii2D;
    float* deformation_accum;
} fdgs_gaussians_out;

int fdgs_densification_stats(void* stream, int N, const int32_t* radii, const uint8_t* visibility_opt,
                             const float* viewspace_grad, int grad_stride, float* max_radii2D_opt,
                             float* xyz_gradient_accum, float* denom);
int fdgs_densify_scratch_bytes(int N, size_t* bytes);
int fdgs_densify_plan(void* stream, int mode, int N, const float* xyz_gradient_accum, const float* denom,
                      const float* scaling, const float* opacity, const float* max_radii2D, const uint8_t* drop_mask,
                      float grad_threshold, float dense_size, float min_opacity, float max_screen_size,
                      float max_world_size, void* scratch, uint32_t* counts_host);
int fdgs_densify_apply(void* stream, const fdgs_gaussians_in* in, const fdgs_gaussians_out* out, const void* scratch,
                       const float* split_samples_opt);

/* ABI 6: row permutation of per-Gaussian arrays of 4-byte elements that share the row index, in one launch:
 * dst[i] = src[perm[i]] (scatter = 0) or dst[perm[i]] = src[i] (scatter = 1), rows of `width` elements.  `perm` is a permutation of 0 .. N-1
 * (int32, device).  The Python host reads an unordered model's parameters through the cached Hilbert permutation of its positions
 * (fdgs.render) and hands the per-Gaussian gradients back through the inverse. */
#define FDGS_MAX_ROW_ARRAYS 8
typedef struct fdgs_row_array { const void* src; void* dst; int width; } fdgs_row_array;
int fdgs_permute_rows(void* stream, int N, const int32_t* perm, int narrays, const fdgs_row_array* arrays, int scatter);

#ifdef __cplusplus
}
#endif
#endif /* FDGS_H */
