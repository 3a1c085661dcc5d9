// Internal (non-ABI) declarations shared by ks_kernels.hip and ks_capi.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace ks {

// Everything a fused-stepper launch needs; passed by value (lands in SGPRs).
struct StepArgs {
    double* u;             // [E,N] fp64 state, in/out
    const float* phi;      // [E,N] fp32 forcing field or nullptr
    const float* actions;  // [E,n_act] fp32 or nullptr
    const float* F;        // [n_act,N] fp32 forcing matrix (needed iff actions)
    const int* env_ids;    // [n_rows] subset or nullptr (identity)
    float* obs;            // [E,N] fp32 out or nullptr (indexed by env id)
    double* ssq_sum;       // [E] out or nullptr (indexed by env id)
    int* status;           // [E] out or nullptr (indexed by env id)
    int n_rows;            // number of envs this launch processes
    int n_act;
    int N;
    long n_substeps;
    // fast mode
    double c_lin[5];       // merged linear stencil: -(D4_k/dx^4 + D2_k/dx^2), k = 0..4
    double mh_inv_dx;      // -0.5 / dx
    double hdt, dt6, dt3;  // dt/2, dt/6, dt/3
    // both modes
    double dt;
    // exact mode: the reference's divisors and their correctly rounded reciprocals (div_const in ks_kernels.hip)
    double dx, dx2, dx4;
    double r_dx, r_dx2, r_dx4;
};

struct Layout {
    int variant;       // ks_variant
    int G;             // lanes per env (0 for LDS variant)
    int P;             // points per lane
    int block;         // threads per workgroup
    int grid;          // workgroups
    size_t lds_bytes;  // dynamic LDS
};

// Returns false if (variant, N) has no instantiated kernel.
bool layout_supported(int variant, int N);
// Launch the fused stepper described by `lay` on `stream`.
hipError_t launch_step(const Layout& lay, int mode, const StepArgs& a, hipStream_t stream);
// rhs test hook: u [n,N], phi [n,N] -> outputs [n,N] (device pointers; ux/uxx/uxxxx may be null)
hipError_t launch_rhs(const double* u, const float* phi, int n_rows, int N, double dx, double dx2,
                      double dx4, double* rhs, double* ux, double* uxx, double* uxxxx,
                      hipStream_t stream);
// cross-lane primitive self-test; d_fail is a device unsigned (bit per ks_variant)
hipError_t launch_selftest(unsigned* d_fail, hipStream_t stream);

}  // namespace ks
