/*
 * sol_hip.h  --  C ABI of libsol_hip.so, the MI355X (gfx950) engine for the
 * solver-in-the-loop hot path.
 *
 * The reference exposes no FFI of its own for this path: the seam is
 * the Python call surface of karman_train.py / burgers_train.py, and the nearest plug
 * point is PhiFlow's PressureSolver / CUDASolver custom op imported at
 * /root/reference/karman-2d/karman_train.py:51.  Each entry point below cites the
 * reference lines whose arithmetic it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; sol_last_error() gives the text
 *     (thread local).  Nothing is allocated inside: all buffers, including workspaces,
 *     are caller-owned DEVICE pointers (fp32 unless noted).
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous on that stream.  The only
 *     process-wide mutable state is the option table behind sol_set_option() (kernel-variant
 *     selection, read at every call) and the launch profiler (sol_prof_begin/end); the library never
 *     reads the environment.  Calls on distinct streams are thread-safe as long as no thread changes
 *     an option or profiles concurrently.
 *   - layouts: density d [B,Y,X]; v_y [B,Y+1,X]; v_x [B,Y,X+1]  (component 0 = y, the
 *     reference's `velocity.data[0]`, karman_train.py:367); images NHWC; conv kernels
 *     HWIO (Keras layout); `params`/`grads` = Keras get_weights() order, flattened.
 */
#ifndef SOL_HIP_H
#define SOL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOL_OK 0
#define SOL_ERR_ARG (-1)      /* invalid argument / unsupported shape */
#define SOL_ERR_HIP (-2)      /* HIP runtime error (text in sol_last_error) */
#define SOL_ERR_WORKSPACE (-3)
#define SOL_ERR_GRAPH (-4)    /* a captured graph holds a node type that is refused (sol_graph_check) */

#define SOL_MARS_MOON_PARAMS 260354   /* model_mars_moon, karman_train.py:101-138 */

const char* sol_last_error(void);
int sol_version(void);
/* sizeof() of the ABI structs below, for bindings that mirror them (ctypes): a stale library is detected at load time */
int sol_abi_sizes(int32_t* karman_cfg, int32_t* burgers_cfg, int32_t* train_cfg);

/* Kernel-variant options (process wide; defaults in parentheses).  The reference's knob of this kind is the
 * `pressure_solver=` argument of KarmanFlow (karman_train.py:167) and the `--cuda` switch (:51); everything else is
 * new.  Unknown names / out-of-range values return SOL_ERR_ARG.
 *   conv_precision (0)  0: fp32-equivalent split arithmetic on the 16-bit matrix pipe (fp16 x3 where the operand's absmax
 *                          is known, bf16 x6 otherwise); 1: bf16 x6 always; 2: strict fp32 MFMA (v_mfma_f32_*_f32) everywhere
 *   cnn_persistent (0)  the ten 32->32 CNN layers of a pass as ONE persistent launch (tagged-granule halo exchange between
 *                       neighbouring workgroups) where the shape allows it (W == 64, ceil(B*H/3) <= #CUs); measured on par
 *                       with the per-layer launches at B = 6
 *   bww_fuse (1), correct_fuse (1), density_mode (0), conv_thin (1), conv_r3 (1), conv_bww32 (1): fusion / kernel choices
 *   k3d_conv_fused (1)  32 -> 32 and 32 -> (<= 16) Conv3D layers (W == 64, operand absmax given) as ONE launch that keeps its accumulators over all
 *                       125 taps (0: five passes of the 2-D kernel with the running sum in HBM); k3d_fused_tf (1): sine transforms
 *                       of the 3-D pressure solve as LDS-resident plane / slab kernels (0: batched GEMMs)
 *   k3d_conv_rows (8)   rows per workgroup of the one-launch Conv3D kernel: 8 = one 64 px x 32 co tile per wave, eight waves (two per
 *                       SIMD), operands prefetched one tap ahead; 6 = 32 x 32 tiles, twelve waves; 3 = 16 x 32 tiles, twelve waves
 *   k3d_tile (0)        1: karman-3d advection from LDS tiles that hold the full z column + halo; 0: wave-per-column gathers
 *                       straight from the L2-resident fields (measured 3x faster at batch 1-2: the tile form is instruction bound)
 *   conv_dx (11)        bit set: 1 the 32 -> 32 fp16 x3 convolutions on the dx-major kernel, 2 thin layers in one-row launches, 4 thin layers
 *                       everywhere, 8 half-channel workgroups in one-row launches
 *   conv_thin_valu (1)  the thin 32 -> (<= 4) layers of 64-pixel images without residual / activation (the trainer's output layer with the
 *                       correction + loss epilogue, the first layer's data gradient) in exact fp32 on the vector ALU; 2: its eight-wave
 *                       form; 0: the split-precision MFMA kernels
 *   seed_fuse (1)       trainer, 64-pixel rows: the loss-gradient seed of an unrolled step (d loss / d v_i + the adjoint of step i+1, scaled
 *                       to the network's output gradient) is computed by the 2 -> 32 backward-data launch in its staging phase; 0: a k_seed
 *                       launch per unrolled step in front of it
 *   fwd_bands (1)       trainer / roll-out, 128 x 64, direct solver, B <= 40: the forward solver step of a simulation as FIVE workgroups in one
 *                       launch (four row bands run the stencil phases with recomputed 8-row halos, a fifth runs the direct solve; hand-offs
 *                       of divergence and pressure through a workspace region); 0: one workgroup per simulation.  Same results bit for bit
 *                       (departure points beyond the halo: recomputed from the step's input, 1e-6)
 *   conv_thin_t3 (1)    thin-input layers (3 -> 32 first layer, 2 -> 32 last backward-data layer incl. the seed form) of 64-pixel rows as three rows
 *                       of the batch x height stack per twelve-wave workgroup; 0: one row per 256-thread workgroup.  Same results bit for bit
 *   k3d_adj_tile (1)    karman-3d: the advection adjoint's fixed-point scatter goes through an int64 LDS window per workgroup (4 x 4 columns + halo 2),
 *                       flushed with one global atomic per non-zero cell; 0: every contribution is a global atomic.  Same results bit for bit
 *   k3d_conv_persist (0) karman-3d: 1 = the one-launch Conv3D kernel as 256 workgroups of consecutive eight-row tiles (the next tile's rows and weight
 *                       sets requested during the last tap rows of the tile) when the tile count is a multiple of 256; same results bit for bit,
 *                       measured slower (register spills), kept as a tested experiment
 *   k3d_bww_jobs (2)    karman-3d: the five depth slices of a 32 -> 32 Conv3D weight gradient (sol_conv3d_bwd_weight*) as ONE launch of ONE round of
 *                       workgroups (51 per slice; other block partition: sums equal to round-off); 1: one launch of five rounds of 32-row
 *                       workgroups (bit-identical to 0); 0: five launches
 *   bww_chunk (0), bww_side (1), streams (1), cpt (0), conv_split3 (0), dbg_skip (0), step_prof (0): experiments, debugging */
int sol_set_option(const char* name, int32_t value);
int sol_get_option(const char* name, int32_t* value);

/* Launch profiler: between sol_prof_begin() and sol_prof_end() every kernel the library launches (eagerly, not under
 * stream capture) carries its own pair of HIP events on the stream it is launched on (hipExtLaunchKernelGGL start/stop:
 * the dispatch's begin/end timestamps, the quantity rocprofv3 --kernel-trace reports).  sol_prof_end synchronises the
 * device and returns the number n of distinct kernels; names (n x 64 chars), total_us[n], calls[n] are HOST arrays. */
int sol_prof_begin(void);
int sol_prof_end(int32_t max_classes, char* names, double* total_us, int32_t* calls);

/* ------------------------------------------------------------------------------------
 * Solver step:  KarmanFlow.step  (karman-2d/karman_train.py:173-185) =
 *   explicit diffusion + velocity BC (lines 175-183) followed by PhiFlow's
 *   IncompressibleFlow.step: semi-Lagrangian advection of density and velocity, inflow,
 *   divergence_free() with the obstacle (hard-BC face masks, divergence, CG pressure
 *   solve, gradient subtraction).  One workgroup per simulation; everything between the
 *   input load and the output store stays in LDS / registers.
 * ---------------------------------------------------------------------------------- */
typedef struct sol_karman_cfg {
    int32_t B, Y, X;        /* batch, cells in y, cells in x (Y%8==0, X in {8,16,32,64}) */
    float dx;               /* cell size = len / X  (karman_train.py:363)              */
    float dt;               /* time step (step(..., dt=1.0))                            */
    float res;              /* `res` argument of step(): alpha = dt*res*res/Re (l.175)  */
    float cg_rtol;          /* CG stops when |r|_2 <= max(cg_rtol*|b|_2, cg_atol)       */
    float cg_atol;
    int32_t cg_max_iter;    /* PhiFlow SparseCG: 2000                                   */
    int32_t grad_pad;       /* 0: replicate (PhiFlow 1.x), 1: dirichlet0                */
    int32_t inflow_before;  /* 0: density += inflow*dt after advection (phi 1.x),
                               1: before advection (karman-2d-phi2/karman_train.py:182) */
    int32_t coarse_n;       /* 0, or (Y/8)*(X/8): size of the coarse space of the CG preconditioner */
    const float* coarse_inv;/* NULL (plain CG), or DEVICE [coarse_n,coarse_n]: inverse of P^T(-A)P for
                               8x8-cell aggregates P of the scene's `active` mask, prepared by the host
                               (see sol_karman_precond_supported).  Only the iteration count changes:
                               the solve still converges to the same tolerance.                      */
    int32_t direct_n;       /* 0, or the number of 32-bit words of `direct`                                   */
    const float* direct;    /* NULL, or DEVICE blob of the DIRECT pressure solver (PhiFlow's PressureSolver plug
                               point, KarmanFlow(pressure_solver=...) karman_train.py:167): fast diagonalisation
                               of the rectangle Laplacian by sine transforms + a dense capacitance correction for
                               the obstacle cells, prepared by the host for the scene's `active` mask (layout:
                               precond.direct_solver_blob).  Takes precedence over the CG; no iteration, the
                               solution equals the converged CG solution up to fp32 round-off.  Built for
                               128 x 64 and for small grids (sol_karman_direct_supported).                     */
} sol_karman_cfg;

/* 1 if the two-level CG preconditioner can be used for a Y x X grid, else 0 */
int sol_karman_precond_supported(int32_t Y, int32_t X);
/* 1 if the direct pressure solver is built for a Y x X grid (128 x 64, and grids of at most 2048 cells with
 * Y % 16 == 0, X in {16, 32, 64}, e.g. the reference's 64 x 32 training recipe), else 0 */
int sol_karman_direct_supported(int32_t Y, int32_t X);

/* Forward step for grids beyond the one-workgroup kernels (the reference generates its data at 256 x 128:
 * karman-2d/karman.py:98-159, Makefile:19-28 `-r 128`).  Same arithmetic and argument meaning as sol_karman_step_fwd,
 * decomposed into chip-wide launches on global memory; the pressure system is solved DIRECTLY, so cfg.direct (blob
 * with a 16/32/64 window, precond.direct_solver_blob(active, max_window=64)) is required.  Forward only: no saved
 * state, no adjoint.  Outputs must not alias inputs.  `direct_header_host`: HOST copy of the first 16 words of the blob
 * (grid, window origin/size, number of perturbed cells: they size the launches).  `workspace`: DEVICE scratch of
 * sol_karman_step_large_workspace_bytes(cfg) bytes. */
size_t sol_karman_step_large_workspace_bytes(const sol_karman_cfg* cfg);
int sol_karman_step_fwd_large(const sol_karman_cfg* cfg, void* stream,
                              const float* d_in, const float* vy_in, const float* vx_in,
                              const float* re, const float* active, const float* inflow,
                              const float* velBCy, const float* velBCyMask, int64_t bc_batch_stride,
                              float* d_out, float* vy_out, float* vx_out,
                              float* feat_out, const float* feat_scale,
                              const int32_t* direct_header_host,
                              void* workspace, size_t workspace_bytes);

/* active  [Y,X]  1 - obstacle mask (cell centres inside Obstacle geometries -> 0)
 * inflow  [Y,X]  inflow rate mask (Inflow(box[5:10,25:75]) -> 1 inside)
 * velBCy, velBCyMask  [Y+1,X] (bc_batch_stride 0) or [B,Y+1,X] (stride (Y+1)*X)
 * saved_vy/saved_vx: post-diffusion+BC velocity kept for the backward pass (NULL: inference)
 * feat_out [B,Y,X,4] (NULL to skip): fused to_feature (karman_train.py:77-86) scaled by
 *   feat_scale[3] = 1/std (l.416-419), a HOST array; channel 3 is zero padding.
 * iters [B]: CG iterations used per simulation (may be NULL).                           */
int sol_karman_step_fwd(const sol_karman_cfg* cfg, void* stream,
                        const float* d_in, const float* vy_in, const float* vx_in,
                        const float* re, const float* active, const float* inflow,
                        const float* velBCy, const float* velBCyMask, int64_t bc_batch_stride,
                        float* d_out, float* vy_out, float* vx_out,
                        float* saved_vy, float* saved_vx,
                        float* feat_out, const float* feat_scale,
                        int32_t* iters);

/* Adjoint of the step w.r.t. its input velocity (density is a passive tracer and has no
 * adjoint: buoyancy_factor=0, karman_train.py:363).  The pressure adjoint is a second CG
 * solve with the same symmetric matrix (PhiFlow's custom gradient).
 * dfeat [B,Y,X,2] (may be NULL): gradient w.r.t. the fused feature channels 0,1; it is
 *   added as g_v_out += feat_scale[c]*dfeat[...,c] before the adjoint runs.              */
int sol_karman_step_bwd(const sol_karman_cfg* cfg, void* stream,
                        const float* saved_vy, const float* saved_vx,
                        const float* re, const float* active,
                        const float* velBCyMask, int64_t bc_batch_stride,
                        const float* g_vy_out, const float* g_vx_out,
                        const float* dfeat, const float* feat_scale,
                        float* g_vy_in, float* g_vx_in,
                        int32_t* iters);

/* ------------------------------------------------------------------------------------
 * Burgers step: BurgersTest.step / step_with_f (burgers/burgers_train.py:182-187) =
 *   semi-Lagrangian self-advection on the periodic staggered grid followed by PhiFlow's
 *   periodic (spectral) diffusion, expressed as two real circulant matrices cy [Y+1,Y+1]
 *   (for v_y rows) ... see sol_burgers_cfg; then v += dt*f.
 * ---------------------------------------------------------------------------------- */
typedef struct sol_burgers_cfg {
    int32_t B, Y, X;        /* cells; staggered arrays v_y [B,Y+1,X], v_x [B,Y,X+1]  (<= 64) */
    float dx, dt;
} sol_burgers_cfg;

/* circ_* are the symmetric circulant matrices of exp(-(2 pi k)^2 * nu*dt), row major:
 * circ_yp1 [Y+1,Y+1], circ_x [X,X] act on v_y; circ_y [Y,Y], circ_xp1 [X+1,X+1] on v_x.
 * f_y/f_x may be NULL (BurgersTest.step).  saved_* keep the input velocity for backward. */
int sol_burgers_step_fwd(const sol_burgers_cfg* cfg, void* stream,
                         const float* vy_in, const float* vx_in,
                         const float* f_y, const float* f_x,
                         const float* circ_yp1, const float* circ_x,
                         const float* circ_y, const float* circ_xp1,
                         float* vy_out, float* vx_out);
int sol_burgers_step_bwd(const sol_burgers_cfg* cfg, void* stream,
                         const float* vy_in, const float* vx_in,
                         const float* circ_yp1, const float* circ_x,
                         const float* circ_y, const float* circ_xp1,
                         const float* g_vy_out, const float* g_vx_out,
                         float* g_vy_in, float* g_vx_in);

/* Forward-only Burgers step for grids beyond the one-workgroup kernels (up to 1024 x 1024): the reference generates its
 * training data at 128 x 128 (burgers/Makefile:19-29, `burgers.py -r 128`; PhiFlow Burgers.step on the CPU there).  Same
 * arguments as sol_burgers_step_fwd plus a workspace of sol_burgers_step_large_workspace_bytes(cfg).  Not differentiable. */
size_t sol_burgers_step_large_workspace_bytes(const sol_burgers_cfg* cfg);
int sol_burgers_step_fwd_large(const sol_burgers_cfg* cfg, void* stream, const float* vy_in, const float* vx_in,
                               const float* f_y, const float* f_x, const float* circ_yp1, const float* circ_x,
                               const float* circ_y, const float* circ_xp1, float* vy_out, float* vx_out,
                               void* workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------------------------
 * 5x5 SAME convolution, NHWC fp32 tensors.  32-input-channel layers with W % 64 == 0 evaluate their fp32 products as
 * exact 16-bit MFMA products of operand splits with fp32 accumulation (option conv_precision; 2 = fp32 MFMA
 * v_mfma_f32_16x16x4_f32 throughout); the thin 3->32 / 32->2 layers and other shapes run on fp32 MFMA.
 * Replaces keras.layers.Conv2D(filters, 5, padding='same') + LeakyReLU / add
 * (karman_train.py:101-138) and its TF gradients.
 * ---------------------------------------------------------------------------------- */
#define SOL_CONV_FWD 0        /* pack for y = conv(x, w)                       */
#define SOL_CONV_BWD_DATA 1   /* pack for dx = conv(dy, flip(w)^T)             */
/* number of floats of a packed weight buffer for (cin, cout, mode) */
size_t sol_conv5x5_packed_floats(int32_t cin, int32_t cout, int32_t mode);
/* w_hwio [5,5,cin,cout] -> packed (zero padded to the MFMA tile shape) */
int sol_conv5x5_pack(void* stream, const float* w_hwio, int32_t cin, int32_t cout,
                     int32_t mode, float* packed);

/* n <= 24 (layer, mode) pack jobs in ONE launch -- same packed layout as n calls of sol_conv5x5_pack (cin[k], cout[k] as there: of the
 * convolution being RUN; packed[k] of sol_conv5x5_packed_floats(cin[k], cout[k], mode[k]) floats).  For hosts that compose the unrolled
 * step themselves (the Python schedules of model_mercury / the Burgers path): one launch per training step instead of ~100.          */
int sol_conv5x5_pack_jobs(void* stream, int32_t n, const float* const* w_hwio, const int32_t* cin, const int32_t* cout,
                          const int32_t* mode, float* const* packed);

#define SOL_EPI_NONE 0        /* y = conv + bias (+ residual)                       */
#define SOL_EPI_LRELU 1       /* y = lrelu(conv + bias (+ residual))                */
#define SOL_EPI_DLRELU 2      /* y = (conv (+ residual)) * lrelu'(act_ref)  (backward) */
/* x [B,H,W,cin]; packed from sol_conv5x5_pack with the same (cin,cout) (for BWD_DATA pass
 * cin = channels of dy, cout = channels of dx, and the weights packed with mode BWD_DATA
 * from the forward layer's (cout_fwd=cin, cin_fwd=cout)); bias [cout] or NULL;
 * residual/act_ref [B,H,W,cout] or NULL; y [B,H,W,cout].  H*W % 64 == 0, W | 64 or 64 | W. */
int sol_conv5x5(void* stream, const float* x, const float* packed, const float* bias,
                const float* residual, const float* act_ref, float* y,
                int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout,
                int32_t epilogue, float slope);

/* sol_conv5x5 with per-tensor absmax bookkeeping: `x_absmax` / `y_absmax` are DEVICE arrays of SOL_ABSMAX_SLOTS
 * uint32 slots (16-byte aligned; one slot per workgroup of a full-chip launch, because same-address atomics serialise
 * in the L2) whose maximum holds the bit pattern of max|x| / max|y| (non-negative floats compare like unsigned integers).
 * y_absmax (or NULL): slots are raised with atomic max by this launch (the caller zeroes them once per tensor).
 * x_absmax (or NULL): when given for a 32-input-channel, W % 64 == 0 convolution, the fp32 products are evaluated as
 * THREE fp16 MFMA products of operands scaled by a power of two derived from the absmax (22-bit operand splits, fp32
 * accumulation; error relative to max|x| max|w| like the fp32 kernel's) instead of six bf16 products.  The producer of x
 * publishes its absmax in the unrolled training graph, so no extra pass over the data exists. */
#define SOL_ABSMAX_SLOTS 256
int32_t sol_absmax_slots(void);   /* == SOL_ABSMAX_SLOTS of the library that was loaded */
/* slots[0..SOL_ABSMAX_SLOTS) = bit patterns whose maximum is max|x| over x[0..n): for tensors whose producer did not publish it */
int sol_absmax(void* stream, const float* x, int64_t n, uint32_t* slots);
int sol_conv5x5_scaled(void* stream, const float* x, const float* packed, const float* bias,
                       const float* residual, const float* act_ref, float* y,
                       int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout,
                       int32_t epilogue, float slope, const uint32_t* x_absmax, uint32_t* y_absmax);

/* dW[5,5,cin,cout] += sum_px x[px+tap] * dz[px];  db[cout] += sum_px dz[px].
 * `partial` is a caller workspace of sol_conv5x5_bwd_weight_ws_floats() floats that the
 * caller zeroes once and may reuse to ACCUMULATE over many calls (the unrolled steps share
 * the weights); sol_conv5x5_bwd_weight_reduce() folds it into dw/db.                     */
size_t sol_conv5x5_bwd_weight_ws_floats(int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout);
int sol_conv5x5_bwd_weight(void* stream, const float* x, const float* dz, float* partial,
                           int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout);
int sol_conv5x5_bwd_weight_reduce(void* stream, const float* partial, float* dw_hwio, float* db,
                                  int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout,
                                  int32_t accumulate);

/* sol_conv5x5_bwd_weight_reduce for n <= 12 layers of one image size (B, H, W) in two launches; same summation order.  dw_hwio[k] /
 * db[k] may point into a flat gradient buffer (Keras get_weights() order).  cin[k] = real input channels of layer k.            */
int sol_conv5x5_bwd_weight_reduce_jobs(void* stream, int32_t n, float* const* partial, float* const* dw_hwio, float* const* db,
                                       int32_t B, int32_t H, int32_t W, const int32_t* cin, const int32_t* cout, int32_t accumulate);

/* ------------------------------------------------------------------------------------
 * Whole training step: the unrolled msteps graph of karman_train.py:397-457
 *   for i in range(msteps): state = simulator_lo.step(state);  state.velocity += CNN(state)
 *   loss = sum_i l2_loss((gt_i - prd_i)/std_v)/msteps;  Adam.
 * ---------------------------------------------------------------------------------- */
typedef struct sol_train_cfg {
    sol_karman_cfg karman;
    int32_t msteps;
    float std_v0, std_v1;   /* dataStats['std'][1]  (karman_train.py:234-255,419,432) */
    float std_re;           /* dataStats['ext.std'][0]                                 */
    float lrelu_slope;      /* Keras LeakyReLU default 0.3                              */
    /* --pretf (karman_train.py:351-355, 416-421): a pre-trained supervised model brings its OWN input / output
     * normalisation ('in.std', 'out.std' of its stats.pickle) while the loss keeps dataStats['std'].  0 = use std_v*. */
    float in_std_v0, in_std_v1;     /* feature scale of the velocity channels: feat = v / in_std       */
    float out_std_v0, out_std_v1;   /* correction scale: velocity += out_std * CNN(feat)                */
} sol_train_cfg;

size_t sol_train_workspace_bytes(const sol_train_cfg* cfg);

/* params/grads: SOL_MARS_MOON_PARAMS floats.  gt_vy [msteps,B,Y+1,X], gt_vx [msteps,B,Y,X+1].
 * Outputs: grads (overwritten), loss_steps[msteps] (device, the per-step l2_loss values,
 * NOT yet divided by msteps), optional prd_* = final corrected state (may be NULL),
 * iters_fwd/iters_bwd [msteps*B] CG iteration counts (device, may be NULL).             */
int sol_train_fwd_bwd(const sol_train_cfg* cfg, void* stream,
                      const float* params,
                      const float* d0, const float* vy0, const float* vx0, const float* re,
                      const float* active, const float* inflow,
                      const float* velBCy, const float* velBCyMask, int64_t bc_batch_stride,
                      const float* gt_vy, const float* gt_vx,
                      void* workspace, size_t workspace_bytes,
                      float* grads, float* loss_steps,
                      float* d_final, float* vy_final, float* vx_final,
                      int32_t* iters_fwd, int32_t* iters_bwd);

/* The same step as a replayable hipGraph: one host call per training step instead of ~1000 kernel
 * launches.  All pointers are baked in at creation (re-create when a buffer moves); new data is
 * fed by copying into the same buffers.  (Option `streams` > 1 splits the batch into chains of
 * simulations on separate internal HIP streams; measured slower on MI355X, default 1.)
 * One training call at a time per process (the chains share an internal stream pool).          */
typedef struct sol_train_graph sol_train_graph;
int sol_train_graph_create(const sol_train_cfg* cfg, const float* params,
                           const float* d0, const float* vy0, const float* vx0, const float* re,
                           const float* active, const float* inflow,
                           const float* velBCy, const float* velBCyMask, int64_t bc_batch_stride,
                           const float* gt_vy, const float* gt_vx,
                           void* workspace, size_t workspace_bytes,
                           float* grads, float* loss_steps,
                           float* d_final, float* vy_final, float* vx_final,
                           int32_t* iters_fwd, int32_t* iters_bwd, sol_train_graph** out);
int sol_train_graph_launch(sol_train_graph* graph, void* stream);
int sol_train_graph_destroy(sol_train_graph* graph);

/* Graph-capture guard.  The reference builds its TF graph once and runs it many times (karman_train.py:385-391, 502); here that shape is a
 * captured hipGraph, and a graph on this path may hold KERNEL nodes only (plus empty / event / child-graph nodes): memset nodes replay
 * unreliably on ROCm 7.2 (a torch reduction's semaphore clear inside a captured trainer made the reported losses 0.5x / 2x the true values
 * after a few replays), memcpy / host / alloc nodes mean work that is not a kernel of this path sits in the replayed region.
 *   sol_graph_census  counts[t] = number of nodes of hipGraphNodeType t in `graph` (a hipGraph_t; child graphs are entered), t < ncounts <= 64
 *   sol_graph_check   SOL_ERR_GRAPH (and a message naming the node types and `what`) when the graph holds a memset, memcpy, memcpy-to/from-
 *                     symbol, host, mem-alloc or mem-free node; SOL_OK otherwise.  sol_train_graph_create applies it to its own capture; a
 *                     host that captures the per-op entry points itself calls it between hipStreamEndCapture and hipGraphInstantiate.    */
/*   sol_copy_words    dst[0..nwords) = src[0..nwords) (32-bit words; src NULL: zero fill) as ONE KERNEL launch: the copy a host uses inside
 *                     a stream capture instead of hipMemcpyAsync / hipMemsetAsync (torch: tensor.copy_ / clone() of a contiguous tensor).   */
int sol_copy_words(void* stream, void* dst, const void* src, int64_t nwords);
/*   sol_clock_probe   measurement aid (bench.py): every SIMD runs a chain of `iters` dependent v_mfma_f32_16x16x16_f16; out4 (device) =
 *                     {duration in 10 ns ticks (s_memrealtime), duration in s_memtime ticks, iters, 0}.  ns per dependent MFMA follows
 *                     the engine clock the device holds under matrix load: it tells a slow box from a slower kernel.                  */
int sol_clock_probe(void* stream, uint64_t* out4, int32_t iters);
/*   sol_clock_stamp   out16 (device, zero it first) [xcd] = {s_memtime, s_memrealtime} of XCD `xcd` (0..7) at this point of the stream (every XCD
 *                     has its own counter).  Two stamps around a region give the AVERAGE shader clock the device held while the region ran:
 *                     100 MHz x d(memtime) / d(memrealtime) per XCD (gfx950: s_memtime counts engine clocks).  bench.py brackets its timed steps.  */
int sol_clock_stamp(void* stream, uint64_t* out16);
/*   sol_latency_probe measurement aid (bench.py): one lane walks `steps` dependent loads through a ZERO-FILLED device buffer of `nlines` 128-byte
 *                     lines (a power of two) in a pseudo-random order; out4 (device) = {duration in 10 ns ticks, steps, last index, -}.  ns per
 *                     load = the memory round trip a kernel's first loads pay (the L2 is invalid at every kernel boundary).                   */
int sol_latency_probe(void* stream, const uint32_t* buf, int64_t nlines, int32_t steps, uint64_t* out4);
int sol_graph_census(void* graph, int32_t* counts, int32_t ncounts);
int sol_graph_check(void* graph, const char* what);
const char* sol_graph_node_type_name(int32_t type);

/* Forward only (karman_apply.py:138-158 roll-out without the frame dump): runs `nsteps`
 * solver+CNN steps in place of d/vy/vx.  workspace: sol_rollout_workspace_bytes().       */
size_t sol_rollout_workspace_bytes(const sol_train_cfg* cfg);
int sol_rollout(const sol_train_cfg* cfg, void* stream, const float* params,
                float* d, float* vy, float* vx, const float* re,
                const float* active, const float* inflow,
                const float* velBCy, const float* velBCyMask, int64_t bc_batch_stride,
                int32_t nsteps, void* workspace, size_t workspace_bytes, int32_t* iters);

/* Stand-alone loss of ONE unrolled step (karman_train.py:428-436: tf.nn.l2_loss((gt.staggered - prd.staggered) / std_v)):
 *   loss (+)= 0.5 * sum_c sum_e ((gt_c[e] - v_c[e]) / std[c])^2        (accumulate_loss = 0 overwrites loss[0])
 *   g_c[e] (+)= gscale * (v_c[e] - gt_c[e]) / std[c]^2                  (d loss / d v; g may be NULL: forward only; gscale = 1/msteps
 *                                                                        for the reference's loss = sum_i / msteps, l.436)
 * v / gt / g / n / std: HOST arrays of ncomp (1..3) entries -- the staggered components v_y [B,Y+1,X], v_x [B,Y,X+1] (, v_z): the
 * zero-padded faces of staggered_tensor() contribute nothing.  scratch: sol_l2_loss_scratch_floats() device floats.  The sum is
 * folded in a fixed order (bit-reproducible).  The trainers (sol_train_*) fuse this loss into their last CNN layer; this entry
 * point serves hosts that compose the step from the per-op ABI. */
int32_t sol_l2_loss_scratch_floats(void);
int sol_l2_loss_fwd_bwd(void* stream, int32_t ncomp, const float* const* v, const float* const* gt, float* const* g,
                        const int64_t* n, const float* std, float gscale, int32_t accumulate_g,
                        float* loss, int32_t accumulate_loss, float* scratch);

/* tf.compat.v1.train.AdamOptimizer update (karman_train.py:449-457), epsilon-hat form:
 *   lr_t = lr*sqrt(1-b2^t)/(1-b1^t);  p -= lr_t*m/(sqrt(v)+eps).
 * clip_norm > 0: per-tensor tf.clip_by_norm(g, clip_norm) first (l.451-454), tensors given
 * by tensor_offsets[n_tensors+1] (HOST array, element offsets).  t is the 1-based step.  */
int sol_adam_tf_step(void* stream, float* params, const float* grads, float* m, float* v,
                     int64_t n, int32_t t, float lr, float beta1, float beta2, float eps,
                     float clip_norm, const int64_t* tensor_offsets, int32_t n_tensors,
                     float* scratch /* >= n_tensors floats, device */);

/* ------------------------------------------------------------------------------------
 * Data-parallel exchange (new capability; the reference is single device, karman_train.py:22,49):
 * the gradients of independent simulations add (the loss is a batch SUM, karman_train.py:430), so N ranks, one per
 * GPU, run the full unroll on their shard and SUM the flat gradient once per training step over RCCL / xGMI.
 * RCCL is bound at run time (dlopen librccl.so.1); nothing here is needed on one GPU.
 *   rank 0: sol_comm_unique_id(id) -> the host distributes the 128 bytes (any side channel) ->
 *   every rank: sol_comm_init(id, nranks, rank, &comm) with ITS device current -> sol_allreduce_grads per step.
 * ---------------------------------------------------------------------------------- */
#define SOL_COMM_ID_BYTES 128
typedef struct sol_comm sol_comm;
int sol_comm_unique_id(char* id /* [SOL_COMM_ID_BYTES], host */);
int sol_comm_init(const char* id, int32_t nranks, int32_t rank, sol_comm** out);
/* in-place SUM over all ranks of flat_grad[count] (device fp32), asynchronous on `stream` */
int sol_allreduce_grads(sol_comm* comm, void* stream, float* flat_grad, int64_t count);
int sol_comm_destroy(sol_comm* comm);

/* ------------------------------------------------------------------------------------
 * karman-3d (BASELINE.json configs[4]: 128 x 64 x 64).  The reference has NO 3-D code (/root/reference/README.md:37-38);
 * these entry points are the dimension-generic twins of the 2-D ones: the same KarmanFlow.step
 * (karman-2d/karman_train.py:173-185) with a third axis, and model_mars_moon (karman_train.py:101-138) with Conv3D(5)
 * layers.  Layout: density [B,Y,X,Z], v_y [B,Y+1,X,Z], v_x [B,Y,X+1,Z], v_z [B,Y,X,Z+1] (y = flow direction, z
 * contiguous); CNN tensors NDHWC = [B,Y,X,Z,C].  Forward step (sol_karman3d_step_fwd), its adjoint w.r.t. the input
 * velocity (sol_karman3d_step_bwd), the Conv3D forward / backward-data (sol_conv3d, incl. the fused SOL_EPI_DLRELU reverse
 * sweep epilogue) and the Conv3D weight gradient (sol_conv3d_bwd_weight): everything karman3d.Karman3DTrainer composes.
 * ---------------------------------------------------------------------------------- */
typedef struct sol_karman3d_cfg {
    int32_t B, Y, X, Z;     /* batch, cells per axis                                                   */
    float dx;               /* cell size = len / X                                                      */
    float dt;               /* time step                                                                */
    float res;              /* `res` of step(): alpha = dt*res*res/Re (karman_train.py:175)             */
    int32_t grad_pad;       /* 0: replicate (PhiFlow 1.x), 1: dirichlet0                                */
    int32_t inflow_before;  /* 0: density += inflow*dt after advection, 1: before                       */
    int32_t direct_n;       /* number of 32-bit words of `direct`                                        */
    const float* direct;    /* DEVICE blob of the direct pressure solver for the scene's `active` mask
                               (three sine-transform matrices, 1/eigenvalues, capacitance matrix of the
                               obstacle cells; layout: precond3d.direct_solver_blob3d).  Required.       */
} sol_karman3d_cfg;
int32_t sol_abi_size_karman3d(void);      /* sizeof(sol_karman3d_cfg) of the library (checked by the ctypes mirror) */

/* Arguments as sol_karman_step_fwd_large with the third component; active / inflow [Y,X,Z]; velBCy / velBCyMask
 * [Y+1,X,Z] (bc_batch_stride 0) or [B,Y+1,X,Z]; feat_out [B,Y,X,Z,4] (or NULL): fused to_feature = the three components
 * at the low faces of every cell and Re, each times feat_scale[c] (HOST array of 4 = 1/std).  direct_header_host: HOST
 * copy of the first 16 words of the blob.  workspace: sol_karman3d_step_workspace_bytes(cfg) bytes of DEVICE scratch.
 * saved_vy/vx/vz (all three, or all NULL): receive the post-diffusion + BC velocity, the only state the adjoint needs.
 * Outputs must not alias inputs. */
size_t sol_karman3d_step_workspace_bytes(const sol_karman3d_cfg* cfg);
int sol_karman3d_step_fwd(const sol_karman3d_cfg* cfg, void* stream,
                          const float* d_in, const float* vy_in, const float* vx_in, const float* vz_in,
                          const float* re, const float* active, const float* inflow,
                          const float* velBCy, const float* velBCyMask, int64_t bc_batch_stride,
                          float* d_out, float* vy_out, float* vx_out, float* vz_out,
                          float* saved_vy, float* saved_vx, float* saved_vz,
                          float* feat_out, const float* feat_scale, const int32_t* direct_header_host,
                          void* workspace, size_t workspace_bytes);
/* Adjoint of the step w.r.t. its input velocity (the density is a passive tracer: buoyancy_factor = 0, karman_train.py:363).
 * saved_v*: the post-diffusion + BC velocity the forward call stored (saved_vy/vx/vz, all three or NULL there).  The
 * pressure adjoint is a second direct solve with the same symmetric matrix (PhiFlow's custom gradient of the CG solve).
 * The advection adjoint scatters in 64-bit fixed point (integer global atomics, power-of-two scale from max|gradient|):
 * the result is reproducible BIT FOR BIT from run to run.
 * workspace: sol_karman3d_step_bwd_workspace_bytes(cfg) bytes, 16-byte aligned. */
size_t sol_karman3d_step_bwd_workspace_bytes(const sol_karman3d_cfg* cfg);
int sol_karman3d_step_bwd(const sol_karman3d_cfg* cfg, void* stream,
                          const float* saved_vy, const float* saved_vx, const float* saved_vz,
                          const float* re, const float* active, const float* velBCyMask, int64_t bc_batch_stride,
                          const float* g_vy_out, const float* g_vx_out, const float* g_vz_out,
                          float* g_vy_in, float* g_vx_in, float* g_vz_in,
                          const int32_t* direct_header_host, void* workspace, size_t workspace_bytes);
/* velocity += s_c * out[..., c] for the three components (to_staggered + add, karman_train.py:88-90, 424-426):
 * out [B,Y,X,Z,cout], cout >= 3; the last face of each component's own axis receives no correction. */
int sol_karman3d_correct(void* stream, const float* out, int32_t cout, float s0, float s1, float s2,
                         float* vy, float* vx, float* vz, int32_t B, int32_t Y, int32_t X, int32_t Z);
/* The reverse-sweep counterparts (one launch each between the CNN's reverse sweep and sol_karman3d_step_bwd):
 * sol_karman3d_correct_bwd: g_v* += gin_v* on every face (the adjoint of the NEXT unrolled step w.r.t. its input; all three NULL: nothing to add) and
 *   d_out4 [B,Y,X,Z,4] = (s0 g_vy, s1 g_vx, s2 g_vz, 0) at the low faces of every cell -- the adjoint of sol_karman3d_correct, zero padded to the four
 *   channels the thin-layer launches read;
 * sol_karman3d_feature_bwd: g_v*[low faces] += f_c dx4[..., c] -- the adjoint of the scaled feature output of sol_karman3d_step_fwd (f_c = 1 / std_in). */
int sol_karman3d_correct_bwd(void* stream, float* g_vy, float* g_vx, float* g_vz, const float* gin_vy, const float* gin_vx, const float* gin_vz,
                             float s0, float s1, float s2, float* d_out4, int32_t B, int32_t Y, int32_t X, int32_t Z);
int sol_karman3d_feature_bwd(void* stream, const float* dx4, float f0, float f1, float f2, float* g_vy, float* g_vx, float* g_vz,
                             int32_t B, int32_t Y, int32_t X, int32_t Z);

/* 5 x 5 x 5 SAME convolution, NDHWC fp32 (keras.layers.Conv3D(filters, 5, padding='same') + bias / LeakyReLU / add).
 * w_dhwio [5,5,5,cin,cout] (Keras layout, D = y).  Layers with cin = 32 and cout = 32 or <= 16 on W == 64 with x_absmax given run as
 * ONE launch that keeps its accumulators over all 125 taps (conv3d_sb.hip; options k3d_conv_fused, k3d_conv_rows); everything
 * else runs as five passes of the 2-D kernels over the (H, W) planes with the running sum in y, the centre slice last (bias,
 * activation, absmax publish).  Same per-product arithmetic as sol_conv5x5 in both forms.
 * x [B,D,H,W,cin] with cin in {4 (zero padded), 32}; residual [B,D,H,W,cout] or NULL is added before the activation;
 * epilogue SOL_EPI_NONE / SOL_EPI_LRELU / SOL_EPI_DLRELU (times LeakyReLU'(act_ref), act_ref [B,D,H,W,cout]: the backward pass'
 * "data gradient (+ skip gradient) times the activation derivative" in one launch; act_ref NULL otherwise); x_absmax / y_absmax
 * as in sol_conv5x5_scaled (may be NULL).  x != y, D >= 3.
 * Backward-data is the same entry point on weights packed with mode SOL_CONV_BWD_DATA (cin = channels of dy = the forward
 * layer's cout, cout = channels of dx = the forward layer's cin; w_dhwio is the FORWARD kernel): dx = conv3d(dy, flip(w)^T).
 * The weight gradient: sol_conv3d_bwd_weight below. */
size_t sol_conv3d_packed_floats(int32_t cin, int32_t cout);
int sol_conv3d_pack(void* stream, const float* w_dhwio, int32_t cin, int32_t cout, int32_t mode, float* packed);
int sol_conv3d(void* stream, const float* x, const float* packed, const float* bias, const float* residual, const float* act_ref, float* y,
               int32_t B, int32_t D, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t epilogue, float slope,
               const uint32_t* x_absmax, uint32_t* y_absmax);

/* Thin-INPUT Conv3D layers (<= 4 -> 32 channels: the first layer of model_mars_moon in 3-D, karman_train.py:103 generalised, and the
 * output layer's data gradient) with the depth taps packed into the channel axis: x'[..][4 s + c] = x[d + s - 2][..][c] is gathered
 * into ws and the layer runs as ONE 2-D 32 -> 32 convolution over the (H, W) planes (25 taps of K = 32 on the 16-bit matrix pipe
 * instead of 125 taps of K = 4 on the fp32 one; per-product arithmetic of sol_conv5x5 with 32 input channels, option conv_precision).
 * x [B,D,H,W,4] (zero padded from cin channels), y [B,D,H,W,32]; packed: sol_conv3d_thin_packed_floats() floats from
 * sol_conv3d_thin_pack (w_dhwio = the FORWARD kernel: [5,5,5,cin,32] for SOL_CONV_FWD, [5,5,5,32,cin] for SOL_CONV_BWD_DATA; cin = the
 * channels the convolution that is RUN reads, 1..4); ws: sol_conv3d_thin_ws_floats() floats of scratch (the gathered tensor + the
 * absmax slots of x, which this call computes itself); bias / act_ref / epilogue / y_absmax as in sol_conv3d. */
size_t sol_conv3d_thin_packed_floats(void);
size_t sol_conv3d_thin_ws_floats(int32_t B, int32_t D, int32_t H, int32_t W);
int sol_conv3d_thin_pack(void* stream, const float* w_dhwio, int32_t cin, int32_t mode, float* packed);
int sol_conv3d_thin(void* stream, const float* x, const float* packed, const float* bias, const float* act_ref, float* y, float* ws,
                    int32_t B, int32_t D, int32_t H, int32_t W, int32_t epilogue, float slope, uint32_t* y_absmax);

/* Thin-OUTPUT Conv3D layers (32 -> cout <= 4 channels: the network's output layer, karman_train.py:137 generalised, and the first layer's data
 * gradient) with the depth taps packed into the OUTPUT channel axis: ONE 2-D 32 -> 32 convolution y'[q][..][4 s + c] over the (H, W) planes into
 * ws, then y[d][..][c] = bias[c] + sum_s y'[d + s - 2][..][4 s + c].  x [B,D,H,W,32] with its absmax slots (or NULL: computed here), y [B,D,H,W,cout];
 * packed: sol_conv3d_thin_packed_floats() floats from sol_conv3d_thin_out_pack (w_dhwio = the FORWARD kernel: [5,5,5,32,cout] for SOL_CONV_FWD,
 * [5,5,5,cout,32] for SOL_CONV_BWD_DATA; cout = the channels the convolution that is RUN writes); ws: sol_conv3d_thin_ws_floats(); no activation. */
int sol_conv3d_thin_out_pack(void* stream, const float* w_dhwio, int32_t cout, int32_t mode, float* packed);
int sol_conv3d_thin_out(void* stream, const float* x, const float* packed, const float* bias, float* y, float* ws,
                        int32_t B, int32_t D, int32_t H, int32_t W, int32_t cout, const uint32_t* x_absmax);

/* Weight gradient of a thin-input layer in the same packing: ONE pass of the 2-D 32 -> 32 fp16 three-product weight-gradient kernel over
 * the gathered tensor instead of five passes of the thin fp32 kernel.  x [B,D,H,W,4], dz [B,D,H,W,32], W == 64; dz_absmax: the slots the
 * producer of dz published, or NULL (computed here); ws: sol_conv3d_thin_ws_floats(); partial: sol_conv3d_thin_bwd_weight_ws_floats();
 * dw_dhwio [5,5,5,cin_real,32], db [32] are written when do_reduce is set; accumulate_partial / do_reduce as in sol_conv3d_bwd_weight_acc. */
size_t sol_conv3d_thin_bwd_weight_ws_floats(int32_t B, int32_t D, int32_t H, int32_t W);
int sol_conv3d_thin_bwd_weight_acc(void* stream, const float* x, const float* dz, const uint32_t* dz_absmax, float* ws, float* partial,
                                   float* dw_dhwio, float* db, int32_t B, int32_t D, int32_t H, int32_t W, int32_t cin_real,
                                   int32_t accumulate_partial, int32_t do_reduce);

/* ... and of a thin-OUTPUT layer (32 -> cout_real <= 4: the network's last layer): dz4 [B,D,H,W,4] (zero padded) is gathered with the opposite
 * depth offsets and dW [5,5,5,32,cout_real], db [cout_real] come from ONE pass of the 32 -> 32 kernel (the five-pass form pads dz to 32
 * channels and runs five full 32 -> 32 passes).  x [B,D,H,W,32] with x_absmax (or NULL: computed here); ws / partial as above. */
int sol_conv3d_thin_out_bwd_weight_acc(void* stream, const float* x, const uint32_t* x_absmax, const float* dz4, float* ws, float* partial,
                                       float* dw_dhwio, float* db, int32_t B, int32_t D, int32_t H, int32_t W, int32_t cout_real,
                                       int32_t accumulate_partial, int32_t do_reduce);

/* Conv3D weight gradient: dw_dhwio [5,5,5,cin_real,cout] = sum_px x[px + tap] * dz[px], db [cout] = sum_px dz[px]; five passes
 * of the 2-D weight-gradient kernels over the shifted plane ranges.  x [B,D,H,W,cin] (cin in {4, 32}, zero padded from
 * cin_real), dz [B,D,H,W,cout] (cout in {2, 32}); x_absmax / dz_absmax (or NULL): absmax slots of the two tensors -> fp16
 * three-product kernels for the 32 -> 32 case.  partial: sol_conv3d_bwd_weight_ws_floats() floats of scratch; db_scratch: 5 * cout floats. */
size_t sol_conv3d_bwd_weight_ws_floats(int32_t B, int32_t D, int32_t H, int32_t W, int32_t cin, int32_t cout);
int sol_conv3d_bwd_weight(void* stream, const float* x, const float* dz, const uint32_t* x_absmax, const uint32_t* dz_absmax,
                          float* partial, float* dw_dhwio, float* db, float* db_scratch,
                          int32_t B, int32_t D, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t cin_real, int32_t cout_real);
/* The same for a trainer that unrolls n steps (karman_train.py:397-457: one gradient per layer over ALL unrolled steps): called n times
 * per layer on ONE partial buffer -- accumulate_partial = 0 on the first call, 1 afterwards -- with do_reduce = 1 on the last call only;
 * dw / db are written when do_reduce is set and then hold the sum over the n calls. */
int sol_conv3d_bwd_weight_acc(void* stream, const float* x, const float* dz, const uint32_t* x_absmax, const uint32_t* dz_absmax,
                              float* partial, float* dw_dhwio, float* db, float* db_scratch,
                              int32_t B, int32_t D, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t cin_real, int32_t cout_real,
                              int32_t accumulate_partial, int32_t do_reduce);

/* offsets (in floats) of layer l's kernel / bias inside the flat mars_moon parameter
 * vector; l in [0,12).  cin/cout may be NULL.                                            */
int sol_mars_moon_layer(int32_t l, int64_t* kernel_off, int64_t* bias_off, int32_t* cin, int32_t* cout);

#ifdef __cplusplus
}
#endif
#endif /* SOL_HIP_H */
