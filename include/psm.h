/* psm.h -- C-ABI of the MI355X-native pressure-surrogate path ("libpsm_hip.so").
 *
 * Drop-in boundary for the surrogate call of the DLPoissonFoam solvers and the
 * Improved_SM evaluators of pauloacs/Solving-Poisson-s-Equation-through-DL-for-
 * CFD-apllications.  Paths below are relative to that repository.
 *
 *   PM  = Thesis_Work/Chapter5/parallelized/test_case/python_module.py
 *   PCI = Thesis_Work/Chapter5/parallelized/DLPoissonSolver/PythonComm_init.H
 *   PC  = Thesis_Work/Chapter5/parallelized/DLPoissonSolver/PythonComm.H
 *   SMD = Improved_SM/deltaU_to_deltaP/source/pressureSM_deltas/SM_call.py
 *   UGP = Improved_SM/U_to_gradP/evaluation/Eval_dual_Dense_onlycil.py
 *
 * Plain C types only: pointers, sizes, int status.  Every function returns
 * PSM_OK (0) or a negative error; the message is available from
 * psm_last_error().  Nothing here aborts the host solver (the reference
 * dereferences NULL on failure: PCI:11-18, log.DL:34-42).
 *
 * Ownership: the caller owns every buffer it passes; the library copies what
 * it needs before returning and never keeps a caller pointer (the reference
 * borrows the solver's buffer through PyArray_SimpleNewFromData, PC:17).
 * Threading: one in-flight call per handle; handles are independent.
 */
#ifndef PSM_H_
#define PSM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSM_ABI_VERSION 4

/* block layout + reassembly variant */
#define PSM_VARIANT_CHAPTER5 0 /* PM:303-332, 373-472 */
#define PSM_VARIANT_DELTAS   1 /* SMD:452-482, 182-365 */
#define PSM_VARIANT_GRADP    2 /* UGP:476-500, 255-369 */

/* scaling of the PCA coefficients around the network */
#define PSM_SCALER_MAX_ABS 0 /* PM:351,365; UGP:525,531; SMD:521-523,536-537 */
#define PSM_SCALER_STD     1 /* SMD:505-512,532-533 */
#define PSM_SCALER_MIN_MAX 2 /* SMD:513-520,534-535 */

/* arithmetic of the PCA contractions and the dense layers */
#define PSM_PRECISION_F32  0 /* exact-f32 MFMA (v_mfma_f32_32x32x2_f32) */
#define PSM_PRECISION_BF16 1 /* bf16 operands (bases, weights, activations), f32 accumulation (v_mfma_f32_32x32x16_bf16) */

#define PSM_OK               0
#define PSM_ERR_ARG         -1 /* invalid argument / shape */
#define PSM_ERR_STATE       -2 /* call order (model incomplete, no plan, ...) */
#define PSM_ERR_HIP         -3 /* a HIP runtime call failed */
#define PSM_ERR_NO_DEVICE   -4 /* no usable gfx950 device */
#define PSM_ERR_UNSUPPORTED -5 /* shape the reference itself cannot process */
#define PSM_ERR_NOMEM       -6
#define PSM_ERR_GEOMETRY    -7 /* a device-pointer solve ran on a grid whose flow-cell pattern is not the bound one */

/* intermediate results readable with psm_read_stage (parity tests) */
#define PSM_STAGE_X_INPUT    0 /* [rows, p_in]   scaled network input (PM:351)      */
#define PSM_STAGE_RES        1 /* [rows, p_out]  network output, inverse-scaled      */
#define PSM_STAGE_BLOCK_PRED 2 /* [rows, S*S*c_out] decoded blocks (PM:365-366)      */
#define PSM_STAGE_OFFSETS    3 /* [cases, c_out, B] per-block corrections (BC_coor)  */
#define PSM_STAGE_SHIFT      4 /* [cases, c_out] global shift (PM:472)               */

/* kernels of one solve, in launch order (psm_profile_solve) */
#define PSM_K_ENCODE   0
#define PSM_K_REDUCE   1
#define PSM_K_MLP      2 /* all dense layers together */
#define PSM_K_DECODE   3
#define PSM_K_STRIPS   4
#define PSM_K_CHAIN    5
#define PSM_K_PASTE    6
#define PSM_K_COUNT    7

typedef struct psm_handle psm_handle;

/* Replaces the module-level constants of PM:103-134,195,303-304 and the
 * Evaluation(...) constructor arguments of SMD:27 / UGP:31. */
typedef struct psm_config {
  int32_t abi_version;   /* PSM_ABI_VERSION */
  int32_t variant;       /* PSM_VARIANT_* */
  int32_t block;         /* S, block edge (128: PM:303, SMD "shape") */
  int32_t overlap;       /* overlap-strip width; 0 = variant default (12 / 32 / 96) */
  int32_t c_in;          /* input channels (3; 4 for pressureSM_Poisson) */
  int32_t c_out;         /* output channels (1; 2 for U_to_gradP) */
  int32_t p_in;          /* retained input PCs  (PM:113, SMD:87) */
  int32_t p_out;         /* retained output PCs (PM:112, SMD:86) */
  int32_t n_dense;       /* Dense layers including the linear head (PM:125-129) */
  int32_t scaler;        /* PSM_SCALER_* */
  int32_t sdf_channel;   /* channel whose non-zero cells are flow cells (2) */
  int32_t device;        /* HIP device ordinal */
  int32_t max_cases;     /* capacity of the case batch (>=1) */
  int32_t strict_degenerate; /* 1: keep NumPy semantics when the last block row
                                duplicates the previous one (NaN field, UGP:340);
                                0: leave that row out */
  int32_t precision;     /* PSM_PRECISION_* */
} psm_config;

/* ---- lifetime ------------------------------------------------------------ */
/* Replaces Py_Initialize + import python_module (PCI:3-18). */
int psm_create(const psm_config* cfg, psm_handle** out);
void psm_destroy(psm_handle* h);
/* Message of the last failure on `h` (or of the last failed psm_create when h is NULL). */
const char* psm_last_error(const psm_handle* h);

/* ---- model artefacts (PM:103-118,168-170; SMD:70-87,507-519) -------------- */
/* sklearn components_[:p] ([p, S*S*c] row-major) and mean_ ([S*S*c]). */
int psm_set_pca(psm_handle* h, const double* comp_in, const double* mean_in,
                const double* comp_out, const double* mean_out);
/* Keras Dense kernel [n_in, n_out] row-major and bias [n_out]; layer in [0, n_dense). */
int psm_set_dense(psm_handle* h, int32_t layer, int32_t n_in, int32_t n_out,
                  const float* kernel, const float* bias);
/* densePCA_attention ('MLP_attention' of utils.py:455-457; Improved_SM/deltaU_to_deltaP/source/pressureSM_deltas/NNs.py:40-72):
 *   x0 = relu(Dense(depth[0])(inputs));  a = LayerNormalization()(MultiHeadAttention(8, 64)(x0[:, None], x0[:, None]))[:, 0]
 *   for i in 1 .. n_layers-1:  x = relu(Dense(depth[i])(a));  a = LayerNormalization()(x + a)
 *   outputs = Dense(PC_p)(a)
 * maps onto n_dense = n_layers + 2 layers of this handle: layer 0 = the first Dense, layer 1 = the attention block
 * (psm_set_attention), layers 2 .. n_layers = the further Dense layers, the last layer = the head; psm_set_layernorm on layers
 * 1 .. n_layers (residual = 0 on layer 1, 1 behind the others).
 * psm_set_attention: the attention runs over a sequence of length 1 (NNs.py:54), so its softmax is exactly 1 and the query / key
 * projections do not enter the result: out = (x . Wv + bv) . Wo + bo.  Wv [d_model, n_heads, value_dim], bv [n_heads, value_dim]
 * (Keras `value` EinsumDense), Wo [n_heads, value_dim, d_model], bo [d_model] (`attention_output`), float32 row-major.  Folded
 * on the host (float64) into one affine layer without activation in Dense slot `layer` (0 < layer < n_dense - 1).
 * psm_set_layernorm: Keras LayerNormalization over the last axis behind hidden layer `layer` (call after that layer was set):
 *   act = (v - mean(v)) * rsqrt(var(v) + epsilon) * gamma + beta,   v = layer output (after its activation) [+ the layer's own
 *   input when residual != 0 -- NNs.py:64 `x + attn_output`; the layer must then be square], biased variance, float32;
 *   n = the layer's n_out, gamma / beta [n], epsilon > 0 (Keras default 1e-3).
 * Both: float32 and bf16 handles (the normalisation itself is float32 in both); the geometry-bound path is kept.  A
 * normalisation whose consumer is another hidden Dense layer costs no launch (that layer's launch applies it to its input rows);
 * the last one runs as its own small launch. */
int psm_set_attention(psm_handle* h, int32_t layer, int32_t d_model, int32_t n_heads, int32_t value_dim,
                      const float* Wv, const float* bv, const float* Wo, const float* bo);
int psm_set_layernorm(psm_handle* h, int32_t layer, int32_t n, const float* gamma, const float* beta, float epsilon,
                      int32_t residual);
/* conv1D_PCA head (NNs.py:75-124; architecture 'conv1D' of utils.py:452-454: 7 layers of 128-64-32-16-32-64-128 filters):
 * n_layers Conv1D layers (kernel_size taps, 'same' padding, ReLU) over the p_in scaled PCA coefficients seen as a sequence
 * [p_in, 1], then Flatten ([p_in, filters] row-major: index p * filters + c) and the Dense layers of psm_set_dense (the
 * reference has exactly one, Dense(PC_p)), the first of which takes p_in * filters inputs.  kernel [kernel_size, c_in,
 * c_out] float32 (Keras Conv1D layout; cross-correlation, (kernel_size - 1) / 2 zeros in front), bias [c_out]; layer in
 * [0, n_layers), c_in of layer 0 is 1.  Call for every layer BEFORE psm_set_dense(layer 0) and psm_plan_grid.  float32
 * handles only; such a model keeps the general (unbound) solve path unless it has a hidden Dense layer. */
int psm_set_conv1d(psm_handle* h, int32_t layer, int32_t n_layers, int32_t kernel_size, int32_t c_in, int32_t c_out,
                   const float* kernel, const float* bias);
/* max_abs: in_a[0] = max_abs_input_PCA, out_a[0] = max_abs_output_PCA (others ignored);
 * std: in_a = mean_in, in_b = std_in, out_a = mean_out, out_b = std_out  ([p_in]/[p_out]);
 * min_max: in_a = min_in, in_b = max_in, out_a = min_out, out_b = max_out. */
int psm_set_scaler(psm_handle* h, const double* in_a, const double* in_b,
                   const double* out_a, const double* out_b);

/* ---- geometry ------------------------------------------------------------ */
/* Fix the uniform grid shape (grid_shape_y/x of PM:216-217) and build the block
 * tables; the grid-native counterpart of init_func (PM:172-247). */
int psm_plan_grid(psm_handle* h, int32_t ny, int32_t nx);
/* Number of blocks per case of the current plan (len(x_list), PM:332). */
int psm_num_blocks(const psm_handle* h);
/* shape[4] = {ny, nx, c_in, c_out} of the current plan. */
int psm_grid_shape(const psm_handle* h, int32_t* shape);

/* Bind the geometry of the planned grid for the following solves -- the drop-in counterpart of the reference's
 * computeOnlyOnce / init_func split (SM_call.py:89-178, python_module.py:172-246: everything that depends on the
 * obstacle only is done once per simulation).  `grid` [ny, nx, c_in] float32 (host, or device memory when
 * on_device != 0): only its SDF channel (cfg.sdf_channel; flow cell = value != 0, SM_call.py:221) is read.
 * Every quantity the block-offset chain needs is a masked sum of decoded values, i.e. linear in the network
 * output with coefficients that depend on the masks and the model only; binding tabulates those coefficients, and
 * single-case solves then take 6 launches instead of 8: the strip means come from dot products with the last
 * hidden activation inside the head layer's launch, and the decode launch runs the offset chain and writes
 * value - offset - shift straight into the field (the decoded blocks are never stored; psm_read_stage(PSM_STAGE_PRED)
 * is stale while bound).  Same results as the unbound path up to float32 summation order.
 * CONTRACT: until psm_unbind_geometry / a new bind / a model or plan change, the SDF channel of every solved grid
 * must have the flow-cell pattern of the bound one; the other channels are free.  Case counts other than the bound
 * one keep the general path.
 * The contract is CHECKED ON THE DEVICE at every bound solve: spare waves of the launch that computes the strip dots
 * compare the pattern of the grid being solved (one ballot per 64 pixels) with the bound one.  On a mismatch the
 * solve's field is NaN everywhere -- never a plausible field of the wrong geometry -- and a flag in mapped pinned
 * memory is raised.  The host-buffer entries (psm_solve_grid, psm_wait_grid, psm_ring_wait) see the flag when the
 * solve has finished, drop the binding, solve the same grid again on the general path and return the correct
 * field with PSM_OK (psm_last_error tells, psm_guard_trips counts); after psm_solve_grid_device the NaN field is
 * what the caller gets and the next psm_synchronize returns PSM_ERR_GEOMETRY (binding dropped).  The riders use
 * CUs the head layer leaves idle; PSM_NO_GUARD=1 in the environment removes them (diagnostic).  bf16 handles: the decode rounds the network output to bf16, so their
 * strip dots are taken from the rounded output in a small launch of their own (7 launches; same rounding points as
 * the general bf16 path).  Grids of more than 64 blocks (the reference's shipped 400 x 3000 case has 104) take a
 * two-launch form of the end: one chain launch, then decode + paste in row chunks (7 launches instead of 9).
 * Returns PSM_ERR_UNSUPPORTED (nothing bound, solves unaffected) for configurations outside these paths: >= 64 block
 * columns, > 128 output components, no hidden layer, last hidden layer wider than 1024. */
int psm_bind_geometry(psm_handle* h, const float* grid, int32_t on_device);
/* The same for a case batch: grids [n_cases, ny, nx, c_in], one geometry per case slot.  Solves with exactly
 * n_cases cases (case i on geometry i) then take 7 launches instead of 9: head + strip dots, one chain launch
 * (a workgroup per case), decode + paste.  Other case counts keep the general path. */
int psm_bind_geometry_cases(psm_handle* h, const float* grids, int32_t n_cases, int32_t on_device);
int psm_unbind_geometry(psm_handle* h);
int psm_geometry_bound(const psm_handle* h);   /* 1 while a geometry is bound */
/* The flow-cell pattern that was bound: mask [bound cases][ny*nx] (1 = SDF channel != 0), for callers that want to check
 * the contract above on their side (cap = bytes available).  PSM_ERR_STATE when nothing is bound. */
int psm_bound_mask(const psm_handle* h, uint8_t* mask, size_t cap);
/* Number of solves so far whose grid was not the bound geometry (each dropped the binding). */
int64_t psm_guard_trips(const psm_handle* h);

/* ---- per-step solve: grid-native counterpart of py_func (PM:249-517) and of
 *      Evaluation.timeStep from block extraction to assemble_prediction
 *      (SMD:452-575, UGP:470-547) ------------------------------------------- */
/* Host buffers.  grid: [n_cases, ny, nx, c_in] float32 NHWC, already normalised
 * as at PM:288-297; fields: [n_cases, ny, nx, c_out].  out_scale: per-case factor
 * applied to the decoded blocks (max_abs_p*U_max^2, SMD:551) or NULL for 1.
 * Synchronous, like py_func. */
int psm_solve_grid(psm_handle* h, const float* grid, int32_t n_cases,
                   const float* out_scale, float* fields);
/* Host buffers, asynchronous: a ring of PSM_RING_SLOTS slots, each with its own pinned host buffers, device buffers,
 * scratch, stream and hipGraph (H2D copy -> kernels -> D2H copy = ONE graph replay per ticket), so the copies of
 * neighbouring tickets run on the DMA engines while another ticket's kernels compute (the reference's py_func is
 * synchronous, PM:249-517; this is the form for a caller that owns several independent cases or time steps, e.g. an
 * ensemble of PISO runs).  A slot must have been waited for before it comes round again: PSM_ERR_STATE if
 * PSM_RING_SLOTS tickets are already in flight.  A geometry bound with psm_bind_geometry applies to the ring as well.
 *
 * Three ways in, fastest first:
 *  (1) zero-copy: psm_ring_acquire hands out the next slot's pinned buffers; the caller packs its grid
 *      [n_cases, ny, nx, c_in] straight into *grid_in (the pack of PythonComm.H:2-9 written to pinned memory instead of
 *      `input_vals`), psm_ring_submit enqueues the ticket, psm_ring_wait returns when *fields_out
 *      [n_cases, ny, nx, c_out] holds the result.  *fields_out stays valid until the ticket that reuses the slot
 *      (PSM_RING_SLOTS tickets later) is submitted.
 *  (2) caller-registered memory: after psm_host_register(range) psm_submit_grid_io DMAs straight from `grid` and,
 *      when `fields` lies in a registered range too, straight into `fields` (psm_wait_grid(h, t, NULL) then only
 *      waits).  The caller keeps the ranges allocated until psm_host_unregister / psm_destroy and must not touch
 *      `grid` before the wait returns.  psm_solve_grid uses registered ranges the same way.
 *  (3) pageable memory: psm_submit_grid copies `grid` into the slot (the caller's buffer is free on return) and
 *      psm_wait_grid copies the field out. */
#define PSM_RING_SLOTS 8
int psm_ring_acquire(psm_handle* h, int64_t* ticket, float** grid_in, float** fields_out);
int psm_ring_submit(psm_handle* h, int64_t ticket, int32_t n_cases, const float* out_scale);
int psm_ring_wait(psm_handle* h, int64_t ticket);
/* Give an acquired ticket back without submitting it (the caller decided not to solve after all): its slot is free
 * again and is handed out when its turn comes round (slots rotate in ticket order).  PSM_ERR_ARG for a ticket that was
 * not acquired or was already submitted. */
int psm_ring_release(psm_handle* h, int64_t ticket);
int psm_host_register(psm_handle* h, void* ptr, size_t bytes);
int psm_host_unregister(psm_handle* h, void* ptr);
/* fields may be NULL (destination given to psm_wait_grid instead). */
int psm_submit_grid_io(psm_handle* h, const float* grid, int32_t n_cases, const float* out_scale, float* fields,
                       int64_t* ticket);
int psm_submit_grid(psm_handle* h, const float* grid, int32_t n_cases,
                    const float* out_scale, int64_t* ticket);
/* fields == NULL: the destination given to psm_submit_grid_io. */
int psm_wait_grid(psm_handle* h, int64_t ticket, float* fields);
/* Device buffers (HIP pointers on cfg.device), asynchronous on `stream`
 * (hipStream_t; NULL = the handle's own stream).  out_scale is a HOST pointer
 * (or NULL) read before return. */
int psm_solve_grid_device(psm_handle* h, const float* d_grid, int32_t n_cases,
                          const float* out_scale, float* d_fields, void* stream);
/* Reassembly alone (assemble_prediction, SMD:182 / UGP:255; correction loop PM:373-472)
 * of caller-supplied decoded blocks block_pred[B, S*S*c_out] for ONE case, host
 * buffers, synchronous.  grid supplies the flow mask (its sdf channel). */
int psm_reassemble(psm_handle* h, const float* grid, const float* block_pred, float* fields);
/* Evaluation only (a8): the label blocks of the planned layout with the per-block mean over the flow cells removed,
 *   y_array[b, ..., c][x_array[b, ..., sdf] != 0] -= mean(same selection)          (SMD:487-488, UGP:509-511)
 * grid [ny, nx, c_in] supplies the flow mask, labels [ny, nx, c_out] the label image(s); blocks_out
 * [B, S*S*c_out] has the layout psm_reassemble takes, so that labels -> psm_label_blocks -> psm_reassemble is the
 * reference's self-check of the assembly ("it should be almost perfect in that case", SMD:577-580; live form
 * UGP:546-547 test_dPdx / test_dPdy).  Host buffers, synchronous. */
int psm_label_blocks(psm_handle* h, const float* grid, const float* labels, float* blocks_out);
/* compute_in_block_error (Improved_SM/deltaU_to_deltaP/source/pressureSM_deltas/utils.py:210-243, called at SM_call.py:555-557):
 * the error of the DECODED BLOCKS of the last solve (case 0; on a bound geometry the blocks are decoded again from the stored
 * network output) against the label blocks, before any reassembly.  grid [ny, nx, c_in] = the grid that was solved (flow
 * cells: SDF channel != 0), labels [ny, nx, c_out] in the network's normalised output units; the label blocks are taken and
 * de-meaned like psm_label_blocks and multiplied by the solve's out_scale (SM_call.py:555: y_array * max_abs_p * U_max_norm^2).
 * Over the flow cells of all blocks, NaN differences left out, norm = max(true) - min(true):
 *   out[0] = mean(pred - true) / norm            (what the reference appends to pred_minus_true_block)
 *   out[1] = mean((pred - true)^2) / norm^2      (pred_minus_true_squared_block)
 *   out[2] = norm (norm_true), out[3] = max(pred) - min(pred) (norm_pred), out[4] = number of differences.
 * float64 accumulation; evaluation only (host buffers, synchronous). */
int psm_block_error(psm_handle* h, const float* grid, const float* labels, double* out);
/* ---- mesh-side entry: the contract of PythonComm_init.H / PythonComm.H ------------
 * psm_set_geometry installs the one-time tables that init_func builds (PM:195-243):
 *   vtx_m2g/wts_m2g [ny*nx,3]  simplices + barycentric weights, mesh -> grid (interp_weights, PM:210)
 *   indices         [ny*nx,2]  (ii, jj) image cell of every grid point (PM:233-241); points that
 *                              are outside the domain carry whatever the caller put there (the
 *                              reference: np.zeros in SMD:161, np.empty in PM:225) and are
 *                              scattered in order like the NumPy fancy assignment (last wins)
 *   sdfunct         [ny*nx]    signed-distance image (PM:227,240)
 *   vtx_g2m/wts_g2m [n_cells,3] grid -> mesh (PM:211); both NULL for a caller that only goes mesh -> grid
 *                              (the offline evaluators, SM_call.py:89-180): psm_solve is then refused
 *   maxs            [4]        max_abs_Ux, max_abs_Uy, max_abs_dist, max_abs_p (PM:109)
 *   normalise_sdf   0: SDF channel as is (PM:292), 1: divided by max_abs_dist (SMD:443)
 *   fill_input      0: interpolate (PM:280), 1: interpolate_fill (SMD:421-423)
 *   wall_threshold  cells whose interpolated SDF is below it keep the previous p (0.05, PM:494)
 * It also fixes the grid shape (psm_plan_grid).  psm_solve additionally requires c_in == 3, c_out == 1. */
int psm_set_geometry(psm_handle* h, int64_t n_cells, int32_t ny, int32_t nx,
                     const int32_t* vtx_m2g, const double* wts_m2g, const int32_t* indices,
                     const double* sdfunct, const int32_t* vtx_g2m, const double* wts_g2m,
                     const double* maxs, int32_t normalise_sdf, int32_t fill_input,
                     double wall_threshold);
/* init_func itself (PM:172-247; called through PythonComm_init.H:53-94 with the solver's cell array and the points of
 * its "top" and "obstacle" patches): builds the tables above IN C++ -- no interpreter, no SciPy -- and installs them
 * like psm_set_geometry.  cells [n,5] float64 = (Ux, Uy, Cx, Cy, p) (Ux decides which grid points are interpolable,
 * PM:231), top [n_top,2], obst [n_obst,2] boundary points; `rank` is accepted for signature compatibility (the MPI
 * gather of PM:179-185 stays with the caller).  psm_set_case supplies what the reference reads from files / hard-codes:
 * maxs[4] (the `maxs` file, PM:106-109), delta (5e-3, PM:195), every (boundary-point stride of the SDF, 10, PM:94-95),
 * wall_threshold (0.05, PM:494); defaults are those values with maxs = 1.
 * Differences from the SciPy-built tables (csrc/psm_geometry.cpp): (a) mesh -> grid: own Delaunay triangulation --
 * identical simplices and weights for points in general position; grid points OUTSIDE the hull of the cell centres
 * take the last simplex of the triangulator's list in both codes, which one that is differs (such points are
 * scattered into image cell (0,0) like the reference's zero-initialised `indices`, SMD:161 -- PM:225 leaves them
 * undefined); (b) grid -> mesh: the lattice is cut along fixed diagonals (every lattice square is cocircular: qhull's
 * choice is not reproducible).  A caller that needs qhull's very tables passes them to psm_set_geometry. */
int psm_set_case(psm_handle* h, const double* maxs, double delta, int32_t every, double wall_threshold);
int psm_init_geometry(psm_handle* h, const double* cells, int64_t n, const double* top, int64_t n_top,
                      const double* obst, int64_t n_obst, int32_t rank);
/* The table builder alone, host only (no GPU, no handle): psm_geometry_shape gives the grid shape (and optionally
 * bounds[4] = rounded x_min, x_max, y_min, y_max, PM:197-201) for sizing; psm_geometry_build fills caller-allocated
 * vtx_m2g/wts_m2g [ny*nx,3], indices [ny*nx,2], sdfunct [ny*nx], vtx_g2m/wts_g2m [n,3]. */
int psm_geometry_shape(const double* cells, int64_t n, double delta, int32_t* ny, int32_t* nx, double* bounds);
int psm_geometry_build(const double* cells, int64_t n, const double* top, int64_t n_top, const double* obst,
                       int64_t n_obst, double delta, int32_t every, int32_t* vtx_m2g, double* wts_m2g,
                       int32_t* indices, double* sdfunct, int32_t* vtx_g2m, double* wts_g2m);
const char* psm_geometry_last_error(void);
/* py_func (PM:249-517, serial form PM1:199-444): cells [n,5] float64 = (Ux, Uy, Cx, Cy, p) as
 * packed at PythonComm.H:2-9 -> p_out [n] float64 as read at PythonComm.H:31-36.  `rank` is
 * accepted for signature compatibility (the MPI funnel, PM:258/511, stays with the caller).
 * Synchronous. */
int psm_solve(psm_handle* h, const double* cells, int64_t n, int32_t rank, double* p_out);
/* The same in two halves, for ONE thread that advances several independent cases (one handle, i.e. one geometry and one
 * stream, per case -- the "ensemble of PISO cases" of a parameter study): psm_solve_begin copies `cells` (or DMAs from
 * the registered array) and enqueues the whole step, psm_solve_end waits for it and delivers p into the `p_out` given
 * to begin.  One step in flight per handle; psm_solve = begin + end. */
int psm_solve_begin(psm_handle* h, const double* cells, int64_t n, int32_t rank, double* p_out);
int psm_solve_end(psm_handle* h);
/* Optional: register the solver's own persistent buffers (the `input_vals` array of PythonComm_init.H:53 lives for the
 * whole run; the output array likewise) so that psm_solve DMAs from / to them directly instead of through the
 * handle's pinned staging copies (saves two host memcpys per step).  cells [n_cells,5] and / or p_out [n_cells] (either
 * may be NULL); the caller guarantees they stay allocated, at the same address, until psm_unpin_buffers, a new
 * psm_set_geometry or psm_destroy.  psm_solve calls with other pointers keep using the staging path.
 * With BOTH arrays registered (and up to 50 000 cells; above that the DMA engine's large-copy rate wins) a step issues no copy at all: the first kernel of the call reads `cells`
 * from the registered pages over PCIe (fully coalesced, taking the U_max partial maxima on the way) and the last one stores p
 * into `p_out` -- up to 9 us off the 74 us per call of the DMA-copy form on a 16 k-cell mesh, depending on the host's PCIe read
 * rates.  The caller must not
 * write `cells` or read `p_out` between psm_solve_begin and psm_solve_end. */
int psm_pin_buffers(psm_handle* h, const double* cells, double* p_out);
int psm_unpin_buffers(psm_handle* h);
/* Generic mesh -> grid step of the evaluators (interpolate_fill + scatter, SM_call.py:419-436,
 * pressureSM_Poisson/SM_call.py:580-600): values [n_cells, k] float64 row-major (k columns of cell
 * data) -> grid_out [ny*nx, k] float64 holding, per image cell, the interpolated value of the grid
 * point NumPy's `grid[...][tuple(indices.T)] = v` leaves there (last writer), 0 where nothing is
 * written; fill != 0: NaN where a barycentric weight is negative (utils.interpolate_fill).  Uses the
 * tables of psm_set_geometry.  Host buffers, synchronous. */
int psm_mesh_to_grid(psm_handle* h, const double* values, int64_t n_cells, int32_t k, int32_t fill, double* grid_out);

/* Input features of the pressureSM_Poisson surrogate (pressureSM_Poisson/SM_call.py:588-711): from the
 * interpolated dimensional grids ux, uy, dux, duy [ny,nx] float64 (zero outside the flow) and the raw
 * signed-distance image sdfunct [ny,nx] (0 inside solids) to the normalised grid image
 * grid_out [ny,nx,4] float32 = (arcsinh-smoothed Poisson source term, dUx/U, dUy/U, sdf), each divided by
 * its max_abs.  np.gradient differences masked at solid neighbours (:602-632), Poisson term (:635),
 * smart_arcsin_smooth_transform (:22-69, :646), NaN -> 0 (:704), rescale (:707-710).
 * params[7] = {L (= phi), U (= U_max_norm), k, max_abs_Poisson_term_1, max_abs_delta_Ux,
 * max_abs_delta_Uy, max_abs_dist}.  Host buffers, synchronous; the result feeds psm_solve_grid of a
 * deltas-variant handle with c_in = 4. */
int psm_poisson_features(psm_handle* h, const double* ux, const double* uy, const double* dux, const double* duy,
                         const double* sdfunct, int32_t ny, int32_t nx, const double* params, float* grid_out);
/* scipy.ndimage.gaussian_filter(field, sigma=(sigma_y, sigma_x), order=0) as used at
 * SM_call.py:353-363 (sigma (10,10) on the assembled field, (50,50) on the deltaU-change
 * weight) and Eval_dual_Dense_onlycil.py:366-367: mode 'reflect', truncate 4.  Host buffers
 * [ny, nx] float32, synchronous; in and out may alias.  Independent of the plan. */
int psm_gaussian_filter(psm_handle* h, const float* in, int32_t ny, int32_t nx, double sigma_y,
                        double sigma_x, float* out);

/* U_to_gradP: integrate the assembled (dp/dx, dp/dy) into p (integrate_field,
 * Eval_dual_Dense_onlycil.py:371-416, and the four-quadrant stitching :597-628).
 * psm_set_integration fixes the geometry: sdfunct [ny*nx] (self.sdfunct[:,:,0], also used by the
 * reference as the per-row "reset" index list), the cut (center_y = 200 and center_x of :599-600)
 * and the grid spacings np.diff(xl)[0], np.diff(yl)[0].  psm_integrate_gradp: gradp [ny,nx,2]
 * -> p [ny,nx], host float32 buffers, synchronous.  Returns PSM_ERR_UNSUPPORTED where the
 * reference itself raises (unequal flow-cell counts at the cut, index list outside a block). */
int psm_set_integration(psm_handle* h, int32_t ny, int32_t nx, const double* sdfunct, int32_t center_y,
                        int32_t center_x, double dx, double dy);
int psm_integrate_gradp(psm_handle* h, const float* gradp, float* p_out);

/* Wait for everything submitted through this handle. */
int psm_synchronize(psm_handle* h);

/* ---- introspection ------------------------------------------------------- */
/* Copy an intermediate of the LAST solve to host memory (float32). */
int psm_read_stage(psm_handle* h, int32_t stage, float* dst, size_t dst_floats);
/* Run one solve with a HIP event pair around every kernel group on the
 * launch stream; ms[PSM_K_COUNT] receives the durations in milliseconds. */
int psm_profile_solve(psm_handle* h, const float* d_grid, int32_t n_cases,
                      float* d_fields, float* ms);
/* Accumulated device time (ms) and launch count of kernel group `k`, measured
 * with HIP events on the launch stream while event timing is enabled.
 * on = 0 disables; on = R >= 1 launches the (idempotent) group R times back to
 * back between the two events of every solve, which amortises the ~2.7 us an
 * event pair adds to a single launch. */
int psm_enable_kernel_timing(psm_handle* h, int32_t kernel, int32_t on);
int psm_get_kernel_timing(psm_handle* h, int32_t kernel, double* total_ms, int64_t* launches);

/* Dispatch-level time of EVERY kernel of the solve path: `steps` solves of d_grid through the launch sequence of
 * psm_solve_grid_device on the handle's stream, each dispatch stamped with its own begin / end by
 * hipExtLaunchKernelGGL (the timestamps rocprofv3 --kernel-trace reads).  names [cap][64] receives the kernel
 * names in first-launch order, total_ms / launches [cap] their accumulated duration and dispatch count,
 * *n_kernels the number of distinct kernels (may exceed cap). */
int psm_time_kernels(psm_handle* h, const float* d_grid, int32_t n_cases, float* d_fields, int32_t steps, char* names,
                     double* total_ms, int64_t* launches, int32_t cap, int32_t* n_kernels);
/* The same pass, reported per kernel as the MEDIAN and the 10th / 90th percentile of its dispatch durations in microseconds
 * (p10_us / p90_us may be NULL): one slow dispatch moves a mean of a few hundred samples, not these.  bench.py's roofline uses it. */
int psm_time_kernels_q(psm_handle* h, const float* d_grid, int32_t n_cases, float* d_fields, int32_t steps, char* names,
                       double* median_us, double* p10_us, double* p90_us, int64_t* launches, int32_t cap, int32_t* n_kernels);
/* Host-buffer throughput measured from a C++ loop on the calling thread (no per-call binding overhead), through the
 * public entries only: `steps` solves of the `n_inputs` host grids [n_inputs][n_cases, ny, nx, c_in] in rotation after
 * `warmup` untimed ones; *seconds = wall time of the timed solves (steady_clock, first submission to last result).
 *   mode 0  psm_solve_grid, pageable caller memory (the synchronous py_func contract)
 *   mode 1  psm_submit_grid / psm_wait_grid ring, pageable caller memory, `depth` tickets in flight
 *   mode 2  psm_submit_grid_io ring on caller-registered memory (grids and result buffers registered for the run)
 *   mode 3  psm_ring_acquire / submit / wait: the slots' pinned buffers hold the inputs (packed before the timed
 *           region, like a solver that writes its fields straight into the slot), results are left in the slots
 * last_fields (optional) receives the result of the last solve [n_cases, ny, nx, c_out]. */
int psm_bench_host(psm_handle* h, const float* grids, int32_t n_inputs, int32_t n_cases, int32_t mode, int32_t depth,
                   int32_t steps, int32_t warmup, double* seconds, float* last_fields);

/* Median elapsed time (ms) of `n` EMPTY HIP event pairs recorded back to back on the launch
 * stream: the cost of the event timing itself, to be subtracted from per-launch event times. */
int psm_event_pair_overhead(psm_handle* h, int32_t n, double* median_ms);

/* ---- host-only helpers (no GPU needed) ----------------------------------- */
/* Block layout of a variant: writes up to `cap` rows of (y0, x0, idx_i, idx_j)
 * into blocks[4*cap]; returns the number of blocks (or <0).  n_x / n_y as at
 * PM:306-307, SMD:461-462, UGP:479-480. */
int psm_layout(int32_t variant, int32_t ny, int32_t nx, int32_t block, int32_t overlap,
               int32_t* blocks, int32_t cap, int32_t* n_x, int32_t* n_y);
/* Which block (index into the layout) supplies each output cell after all
 * pastes, and from which local cell: owner[ny*nx] = b*S*S + r*S + c (or -1). */
int psm_owner_map(int32_t variant, int32_t ny, int32_t nx, int32_t block, int32_t overlap,
                  int32_t strict_degenerate, int32_t* owner);
/* Host replay of the device reassembly (strip table -> offset chain -> owner-map
 * paste) on caller-supplied decoded blocks pred[B, S*S*c_out] and one grid
 * [ny, nx, c_in].  Verification helper for the plan tables; never called by
 * psm_solve_*.  offsets [c_out, B] and shifts [c_out] may be NULL. */
int psm_debug_reassemble_host(int32_t variant, int32_t ny, int32_t nx, int32_t block, int32_t overlap,
                              int32_t strict_degenerate, int32_t c_in, int32_t c_out, int32_t sdf_channel,
                              const float* grid, const float* pred, float* fields, float* offsets,
                              float* shifts);
/* Diagnostic: with PSM_GUARD_PAGES=1 in the environment (read once, at the first allocation) every device buffer of the
 * library is placed at the end of its own mapping with an unmapped granule behind it, so that a kernel running past the end
 * of a buffer takes a GPU page fault instead of touching a neighbour (the GPU address sanitizer's stand-in for that class;
 * csrc/psm_alloc.cpp).  psm_debug_guard_pages: 1 when the mode is on; psm_debug_malloc / psm_debug_free: the same allocator
 * for a caller's own test buffers (hipError_t values). */
int psm_debug_guard_pages(void);
int psm_debug_malloc(void** ptr, size_t bytes);
int psm_debug_free(void* ptr);
/* Synchronous copies between device memory and ordinary (pageable) host memory through the library's pinned bounce buffer --
 * what every entry of this library that takes host pointers does internally (the GPU never touches caller memory that was
 * not registered explicitly; csrc/psm_alloc.h says why).  For a caller's own test buffers; hipError_t values. */
int psm_debug_copy_to_device(void* dst_device, const void* src_host, size_t bytes);
int psm_debug_copy_to_host(void* dst_host, const void* src_device, size_t bytes);
int psm_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PSM_H_ */
