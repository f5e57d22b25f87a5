/*
 * mcpc_host.c -- a plain C host of libmcpc.so: no Python, no torch.
 *
 * What a non-Python caller of the hot path looks like: device buffers from the HIP runtime's C
 * API, the network and the call described by the two structs of include/mcpc.h, one MCPC learning call (Langevin steps with
 * the fused Philox kick, Hebbian sums over the sampling steps), the gradient bucket read out and summed over the shards
 * with the library's own RCCL path (a communicator of ONE rank here: the identity), everything written to a file that
 * tests/test_gpu_c_host.py compares with the NumPy oracle.
 *
 *   mcpc_host case.bin out.bin
 *
 * case.bin (little endian): int32 magic 0x4d435043, n_latent, n_in, n_out, batch, T, acc_begin, loss_kind, acts[6], sizes[6];
 *   double lr, noise_var, loss_var (Python floats are doubles: the library rounds them where torch does); uint64 seed; then float32 arrays: W_j [out_j][in_j], b_j [out_j] for every Linear,
 *   target [batch][n_out] (if n_out > 0), x0_l [batch][n_l] for every latent layer.
 * out.bin: double energies [T][8]; float x_l [batch][n_l] per layer; float grads [param_count] (normalised by n_acc * batch).
 *
 * Build (done by __graft_entry__.build()):
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_host/mcpc_host.c \
 *       -Lmontecarlopredictivecoding_amd -lmcpc -L/opt/rocm/lib -lamdhip64 -o examples/c_host/mcpc_host
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mcpc.h"

#define HIP_OK(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); exit(2); } \
    } while (0)
#define MCPC_OK_OR_DIE(call)                                                                      \
    do {                                                                                          \
        int r_ = (call);                                                                          \
        if (r_ != MCPC_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, r_, mcpc_last_error()); exit(3); } \
    } while (0)

static void read_exact(void* dst, size_t bytes, FILE* f) {
    if (bytes && fread(dst, 1, bytes, f) != bytes) { fprintf(stderr, "case file too short\n"); exit(4); }
}

/* host array from the file -> device */
static float* upload(size_t n, FILE* f) {
    float* h = (float*)malloc(n * sizeof(float) + 4);
    float* d = NULL;
    read_exact(h, n * sizeof(float), f);
    HIP_OK(hipMalloc((void**)&d, n * sizeof(float) + 4));
    HIP_OK(hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice));
    free(h);
    return d;
}

static void download(const void* d, size_t bytes, FILE* f) {
    void* h = malloc(bytes + 4);
    HIP_OK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
    if (fwrite(h, 1, bytes, f) != bytes) { fprintf(stderr, "short write\n"); exit(5); }
    free(h);
}

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s case.bin out.bin\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int32_t hdr[8], acts[MCPC_MAX_LATENT], sizes[MCPC_MAX_LATENT];
    double fl[3];
    uint64_t seed;
    read_exact(hdr, sizeof hdr, f); read_exact(acts, sizeof acts, f); read_exact(sizes, sizeof sizes, f);
    read_exact(fl, sizeof fl, f); read_exact(&seed, sizeof seed, f);
    if (hdr[0] != 0x4d435043) { fprintf(stderr, "bad magic\n"); return 1; }
    const int L = hdr[1], n_in = hdr[2], n_out = hdr[3], B = hdr[4], T = hdr[5], acc_begin = hdr[6], loss_kind = hdr[7];
    if (mcpc_abi_version() != MCPC_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    HIP_OK(hipSetDevice(0));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    mcpc_net_desc nd;
    memset(&nd, 0, sizeof nd);
    nd.abi_version = MCPC_ABI_VERSION; nd.n_latent = L; nd.n_in = n_in; nd.n_out = n_out; nd.batch = B; nd.device = 0;
    for (int l = 0; l < L; ++l) { nd.sizes[l] = sizes[l]; nd.acts[l] = acts[l]; nd.ecoef[l] = 1.0f; }
    mcpc_engine* e = NULL;
    MCPC_OK_OR_DIE(mcpc_create(&nd, &e));

    const int n_lin = L + (n_out > 0 ? 1 : 0);
    float *W[MCPC_MAX_LATENT + 1], *b[MCPC_MAX_LATENT + 1];
    for (int j = 0; j < n_lin; ++j) {
        const size_t in = j == 0 ? (size_t)n_in : (size_t)sizes[j - 1], out = j < L ? (size_t)sizes[j] : (size_t)n_out;
        W[j] = upload(out * in, f);
        b[j] = upload(out, f);
        MCPC_OK_OR_DIE(mcpc_bind_params(e, j, W[j], b[j]));
    }
    MCPC_OK_OR_DIE(mcpc_params_changed(e, stream));
    MCPC_OK_OR_DIE(mcpc_bind_inputs(e, NULL, stream));                       /* pseudo-input of zeros */
    float* y = NULL;
    if (n_out > 0) { y = upload((size_t)B * n_out, f); MCPC_OK_OR_DIE(mcpc_bind_target(e, y, stream)); }
    float* x[MCPC_MAX_LATENT];
    for (int l = 0; l < L; ++l) x[l] = upload((size_t)B * sizes[l], f);
    fclose(f);
    MCPC_OK_OR_DIE(mcpc_load_state(e, (const float* const*)x, stream));

    double* energies = NULL;
    HIP_OK(hipMalloc((void**)&energies, (size_t)T * (MCPC_MAX_LATENT + 2) * sizeof(double)));
    HIP_OK(hipMemsetAsync(energies, 0, (size_t)T * (MCPC_MAX_LATENT + 2) * sizeof(double), stream));
    mcpc_run_desc rd;
    memset(&rd, 0, sizeof rd);
    rd.T = T; rd.t_begin = 0; rd.n_steps = T;
    rd.loss_kind = loss_kind; rd.loss_var = fl[2];
    rd.xopt_kind = MCPC_XOPT_SGD; rd.lr = fl[0]; rd.beta1 = 0.9; rd.beta2 = 0.999; rd.eps = 1e-8;
    rd.update_x = 1;
    rd.noise_mode = MCPC_NOISE_PHILOX; rd.noise_var = fl[1]; rd.seed = seed; rd.step_base = 0; rd.chain_base = 0;
    rd.acc_begin = acc_begin; rd.acc_end = T; rd.acc_reset = 1;
    rd.energy_mode = MCPC_ENERGY_ALL; rd.energies_out = energies;
    MCPC_OK_OR_DIE(mcpc_run(e, &rd, stream));
    MCPC_OK_OR_DIE(mcpc_store_state(e, x, stream));

    /* the gradient bucket: normalised as pc_trainer.py:905-909 does, then the one collective of a learning call */
    const int64_t n_par = mcpc_param_count(e);
    float* flat = NULL;
    HIP_OK(hipMalloc((void**)&flat, (size_t)n_par * sizeof(float)));
    MCPC_OK_OR_DIE(mcpc_read_param_grads_flat(e, flat, n_par, 1.0f / ((float)(T - acc_begin) * (float)B), stream));
    unsigned char id[MCPC_COMM_ID_BYTES];
    MCPC_OK_OR_DIE(mcpc_comm_unique_id(id));
    MCPC_OK_OR_DIE(mcpc_comm_init(e, 1, 0, id));
    MCPC_OK_OR_DIE(mcpc_allreduce_grads(e, flat, n_par, stream));
    MCPC_OK_OR_DIE(mcpc_sync_check(e, stream));

    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 1; }
    download(energies, (size_t)T * (MCPC_MAX_LATENT + 2) * sizeof(double), o);
    for (int l = 0; l < L; ++l) download(x[l], (size_t)B * sizes[l] * sizeof(float), o);
    download(flat, (size_t)n_par * sizeof(float), o);
    fclose(o);

    int32_t lds = 0, cpw = 0, nwg = 0, slots = 0;
    MCPC_OK_OR_DIE(mcpc_query(e, &lds, &cpw, &nwg, &slots));
    printf("mcpc_host: %s, %d workgroups of %d chains, %d bytes of LDS, %lld parameters, %d steps\n", mcpc_step_kernel_name(e), nwg, cpw,
           lds, (long long)n_par, T);
    MCPC_OK_OR_DIE(mcpc_destroy(e));
    for (int j = 0; j < n_lin; ++j) { HIP_OK(hipFree(W[j])); HIP_OK(hipFree(b[j])); }
    for (int l = 0; l < L; ++l) HIP_OK(hipFree(x[l]));
    if (y) HIP_OK(hipFree(y));
    HIP_OK(hipFree(flat)); HIP_OK(hipFree(energies));
    HIP_OK(hipStreamDestroy(stream));
    return 0;
}
