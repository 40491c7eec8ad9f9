/* psm_unet.h -- C-ABI of the convolutional surrogate path (libpsm_hip.so).
 *
 * The project's north star names a CNN / U-Net forward pass (Conv2D + bias + ReLU, 2x2 max-pool, 2x
 * nearest-neighbour upsample, skip concatenation, 1x1 head).  The reference repository contains NO such
 * network -- its CNN folders are empty placeholders (Thesis_Work/Chapter4/README.md:3) and the model named
 * "U-Net" at Thesis_Work/Chapter5/parallelized/test_case/python_module.py:131 is a Dense stack -- so there is
 * no reference interface to cite line by line and PARITY IS UNPINNED: the network is the build-defined "UNet-S"
 * of SURVEY.md §8 (row a-conv) / oracle/unet_oracle.py.  What the entry points keep from the reference is the
 * calling convention of its Keras models: NHWC float32 images in, NHWC float32 images out, kernels in Keras'
 * Conv2D layout [kh, kw, c_in, c_out] with 'same' zero padding (the layout `model.load_weights` /
 * `load_model` would deliver, python_module.py:170, SM_call.py:74-79), normalised grid images exactly like
 * those psm_solve_grid takes (python_module.py:288-297).
 *
 * Network: levels l = 0..L-1 with widths w_l; encoder level = [2x2 max-pool] conv3x3+ReLU conv3x3+ReLU;
 * decoder level = concat(upsample2x(below), encoder_l) conv3x3+ReLU conv3x3+ReLU; head = conv1x1, linear.
 * Convolution index order (psm_unet_conv_shape / psm_unet_set_conv): enc0a, enc0b, enc1a, ... enc{L-1}b,
 * dec{L-2}a, dec{L-2}b, ... dec0a, dec0b, head.  Status codes and error text as in psm.h. */
#ifndef PSM_UNET_H
#define PSM_UNET_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct psm_unet psm_unet;

/* widths[n_levels]: channel counts per level (multiples of 16, <= 1024); c_in 1..16, c_out 1..16. */
int psm_unet_create(int32_t c_in, int32_t c_out, int32_t n_levels, const int32_t* widths, int32_t device, psm_unet** out);
void psm_unet_destroy(psm_unet* u);
const char* psm_unet_last_error(const psm_unet* u);
int psm_unet_num_convs(const psm_unet* u);
/* kernel edge (3 or 1), input and output channels of convolution `idx`. */
int psm_unet_conv_shape(const psm_unet* u, int32_t idx, int32_t* k, int32_t* c_in, int32_t* c_out);
/* weight [k, k, c_in, c_out] float32 (Keras Conv2D kernel), bias [c_out]. */
int psm_unet_set_conv(psm_unet* u, int32_t idx, const float* weight, const float* bias);
/* PSM_PRECISION_F32 (default; exact f32 products) or PSM_PRECISION_BF16 (activations and weights rounded to bf16
 * at every convolution input, f32 accumulation by v_mfma_f32_16x16x32_bf16; finished activations are kept in
 * memory as bf16 -- the value the next convolution would round them to anyway --, partial sums of split layers
 * as float32).  Call before psm_unet_plan.  Constants from psm.h. */
int psm_unet_set_precision(psm_unet* u, int32_t precision);
/* bf16 mode fuses the two convolutions of a wide level into one launch and keeps the first one's activation (and the
 * last 3x3 layer's, whose 1x1 head is fused) on chip.  on != 0: those activations are stored as well, so that
 * psm_unet_read_activation can return every layer (parity tests); off by default.  Call before psm_unet_plan. */
int psm_unet_keep_activations(psm_unet* u, int32_t on);
/* Fixes the image size (ny, nx multiples of 2^(n_levels-1)) and the largest case batch; allocates activations. */
int psm_unet_plan(psm_unet* u, int32_t ny, int32_t nx, int32_t max_cases);
/* Host buffers: grid [n, ny, nx, c_in] -> field [n, ny, nx, c_out], synchronous. */
int psm_unet_forward(psm_unet* u, const float* grid, int32_t n_cases, float* field);
/* Device buffers, asynchronous on `stream` (hipStream_t; NULL = the handle's stream). */
int psm_unet_forward_device(psm_unet* u, const float* d_grid, int32_t n_cases, float* d_field, void* stream);
int psm_unet_synchronize(psm_unet* u);
/* Output activation of convolution `idx` of the last forward pass (introspection for parity tests):
 * dst [n_cases * (ny >> level) * (nx >> level) * c_out] floats.  PSM_ERR_STATE for an activation that a fused pair
 * keeps on chip (see psm_unet_keep_activations). */
int psm_unet_read_activation(psm_unet* u, int32_t idx, float* dst, int64_t dst_floats);
/* One forward pass with a HIP event between the layers: ms [num_convs] (each includes ~3 us of event overhead),
 * wgs [num_convs] workgroups launched per layer (may be NULL).  Introspection for tuning. */
int psm_unet_profile(psm_unet* u, const float* d_grid, int32_t n_cases, float* d_field, float* ms, int32_t* wgs);
/* Plan-time autotune (call after psm_unet_plan, before the forward passes that matter).  The planner decides by rules --
 * fuse a level's two convolutions when it has enough tiles, pick the tile shape that fills the chip, split the input channels
 * of layers that cannot fill it -- whose pay-off depends on the shapes of BOTH a layer and its consumer (a split shortens a
 * layer but makes its consumer sum float32 partial-sum slabs).  This measures instead: per level the other pair choice, per
 * unfused layer the two other tile shapes, per split layer a shallower split, each kept when the WHOLE forward pass of
 * n_cases cases gets more than 1 % faster (a few thousand forward passes on zero images in all; the handle's buffers are
 * re-allocated).  us_before / us_after (may be NULL): microseconds per forward pass.  Results are unchanged by the choices
 * up to float32 summation order (bf16: rounding flips).
 * psm_unet_ksplit: the split of convolution idx in the current plan; psm_unet_plan_info: info[4] = {tile rows, channel tiles
 * per workgroup, split, role}: role & 3 = pair role (0 none, 1 leader, 2 computed by the leader's launch), role & 4 = the layer
 * runs the x6 form, role & 8 = in-workgroup K split (bf16 mode: eight-wave workgroups, the two halves of a workgroup's channel
 * chunks on waves 0-3 / 4-7, summed through LDS -- chosen by rule for layers of four or more chunks whose launch has at most
 * one workgroup per CU; PSM_UNET_KW=0 switches it off).
 * x6 (float32 mode, default on; PSM_UNET_X6=0 switches it off): layers with at least 64 input channels and an 8-row tile run
 * their float32 contractions on the bf16 matrix pipe -- activations and weights split EXACTLY into three bf16 planes
 * (x = hi + mid + lo), six MFMA terms per product (hh, hm, mh, hl, lh, mm; the dropped terms are below 2^-24 of the
 * product), float32 accumulation: float32 accuracy (same parity tests and tolerances as the float32 MFMA form) at 6 x 16
 * cycles per 32 channels instead of 8 x 32.  psm_unet_autotune also measures the other choice per layer.  Finite values only:
 * an infinite activation or weight splits into (inf, NaN, NaN), so where the float32 MFMA would return +-inf an x6 layer
 * returns NaN (NaN stays NaN); magnitudes below 2^-110 lose the low planes to underflow. */
int psm_unet_autotune(psm_unet* u, int32_t n_cases, int32_t iters, float* us_before, float* us_after);
/* The planner's per-layer choices as psm_unet_autotune left them, so that a later process can REPLAY the plan instead of
 * measuring again (the 1 % keep rule on measured medians makes the chosen plan run-dependent, and with it the float32
 * summation order): choices [4 * num_convs] = per convolution {split-K cap 1..8, tile shape -1 (rule) / 0..2, pair -1 (rule)
 * / 0 / 1, x6 -1 (rule) / 0 / 1}.  get: returns num_convs (n = capacity of choices, >= 4 * num_convs); set: n must be
 * 4 * num_convs, re-plans at the planned size.  Two handles with the same weights, size and choices produce bit-identical
 * fields. */
int psm_unet_get_choices(const psm_unet* u, int32_t* choices, int32_t n);
int psm_unet_set_choices(psm_unet* u, const int32_t* choices, int32_t n);
int psm_unet_ksplit(const psm_unet* u, int32_t idx);
int psm_unet_plan_info(const psm_unet* u, int32_t idx, int32_t* info);
/* Dispatch-level time of every launch of the forward pass: `steps` passes on the handle's stream, each dispatch stamped with
 * its own begin / end by hipExtLaunchKernelGGL (the timestamps rocprofv3 --kernel-trace reads; no marker packets between
 * the layers).  us [num_convs]: average duration in microseconds of the launch that STARTS at convolution i -- a fused pair
 * or a fused head is one launch, the convolutions it also computes report 0; launches [num_convs] (may be NULL): launches
 * per pass (0 or 1); names [num_convs][64] (may be NULL): the kernel instantiation that ran. */
int psm_unet_time_kernels(psm_unet* u, const float* d_grid, int32_t n_cases, float* d_field, int32_t steps, double* us,
                          int32_t* launches, char* names);
/* The same pass, per launch the MEDIAN and the 10th / 90th percentile (may be NULL) of its dispatch durations in microseconds. */
int psm_unet_time_kernels_q(psm_unet* u, const float* d_grid, int32_t n_cases, float* d_field, int32_t steps, double* median_us,
                            double* p10_us, double* p90_us, int32_t* launches, char* names);
/* Diagnostic builds (-DPSM_STAMPS) only: runs the network up to and including convolution `idx` on the staged input of
 * the last psm_unet_forward and returns workgroup-0 time stamps of that last layer, stamps_us[64] in microseconds
 * after the first (-1: not reached; all -1 in the shipped library). */
int psm_unet_debug_run_layer(psm_unet* u, int32_t idx, float* stamps_us);
/* Algorithmic work of one forward pass of one case at the planned size. */
int64_t psm_unet_flops(const psm_unet* u);

#ifdef __cplusplus
}
#endif
#endif
