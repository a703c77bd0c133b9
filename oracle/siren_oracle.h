/* oracle/siren_oracle.h — CPU restatement of BRIEF's SIREN fit/decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / the timed CPU
 * baseline.  The product (brief_pytorch_amd) never links, imports or calls it.
 *
 * Parity pin: every function here is checked against golden vectors produced by running
 * the reference itself (tests/golden/make_golden.py, tests/test_oracle_golden.py).
 */
#ifndef SIREN_ORACLE_H
#define SIREN_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int cin;          /* coords_channel (2|3)            utils/Networks.py:246 */
    int cout;         /* data_channel  (1|3) */
    int layers;       /* number of Linear layers (>=3 here; first + hidden + head) */
    int features;     /* hidden width F */
    float w0_first;   /* YAML Module.phi.w0 */
    float w0_hidden;  /* hard-coded 30 in the reference (Sine() default) */
    int output_act;   /* head followed by Sine() */
} oracle_desc;

/* utils/Networks.py:292-297 calc_param_count (res=False) */
size_t oracle_param_count(const oracle_desc *d);
int oracle_num_threads(void);
void oracle_set_num_threads(int n);

/* torch.linspace(lo,hi,n) as torch-CPU computes it (utils/dataset.py:28-32) */
void oracle_linspace(float lo, float hi, int64_t n, float *out);
/* flattened grid coords for linear voxel indices (utils/dataset.py:36-60, order (d,h,w));
 * dims[ndim], idx==NULL means 0..n-1 */
void oracle_grid_coords(const int64_t *dims, int ndim, float lo, float hi, const int64_t *idx, int64_t n, float *out);

void oracle_forward_f32(const oracle_desc *d, const float *params, const float *x, int64_t n, float *yhat);
void oracle_forward_f64(const oracle_desc *d, const float *params, const float *x, int64_t n, float *yhat);
/* loss_kind: 0 datal2, 1 datasmoothl1 (main.py:176-191) */
double oracle_loss_grad_f32(const oracle_desc *d, const float *params, const float *x, const float *y, const float *w,
                            int64_t n, int loss_kind, float thr, float beta, float *grads, float *yhat_out, float *w_eff_out);
double oracle_loss_grad_f64(const oracle_desc *d, const float *params, const float *x, const float *y, const float *w,
                            int64_t n, int loss_kind, float thr, float beta, float *grads, float *yhat_out, float *w_eff_out);

/* torch.optim single-tensor updates (utils/misc.py:174-183): kind 0 Adamax, 1 Adam, 2 SGD; t is 1-based */
void oracle_optim_step(int kind, float *p, const float *g, float *s1, float *s2, int64_t n,
                       double lr, double b1, double b2, double eps, int64_t t);

/* utils/io.py:136-147 invnormalize_data('minmaxany_a_b') followed by the truncating cast;
 * dtype_code: 0 uint8, 1 uint16 ; out is written as that type */
void oracle_invnormalize(const float *yhat, int64_t n, float scale_min, float scale_max, double vmin, double vmax,
                         int dtype_code, void *out);
/* utils/io.py:65-80 normalize_data('minmaxany_a_b') for uint8/uint16 sources */
void oracle_normalize(const void *src, int dtype_code, int64_t n, float scale_min, float scale_max, double vmin, double vmax, float *out);

#ifdef __cplusplus
}
#endif
#endif
