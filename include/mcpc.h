/*
 * mcpc.h -- C ABI of libmcpc.so: the MI355X-native Monte Carlo Predictive Coding engine.
 *
 * Drop-in boundary for ONE hot path of gaspardol/MonteCarloPredictiveCoding: the body of
 * PCTrainer.train_on_batch's `for t in range(T)` loop
 * (reference predictive_coding/pc_trainer.py:712-981) together with the Langevin callback
 * random_step (reference utils/model.py:35-44), for networks of the shape built by
 * get_model (reference utils/model.py:47-69) and by the toy scripts
 * (reference figure_2.py:40-44, figure_3.py:50-55):
 *
 *     Sequential[ Linear, PCLayer, (act), Linear, PCLayer, (act), ..., Linear (, PCLayer) ]
 *
 * The reference has no FFI layer of its own (it is pure Python on torch); the entry points
 * below are what a binding for this path has to call, each annotated with the reference
 * lines it replaces.  Plain pointers and sizes only: no torch types.  All `float*`/`double*`
 * arguments are DEVICE pointers (HIP, gfx950) unless stated otherwise; `stream` is a
 * hipStream_t passed as void*.  Every function returns 0 on success or a negative MCPC_E*
 * code; mcpc_last_error() returns a human-readable message for the calling thread.
 *
 * Threading: one engine per (device, stream); an engine is not re-entrant.  All launches are
 * asynchronous on the given stream; the calls that wait for the device are mcpc_create / mcpc_destroy,
 * mcpc_sync_check and the two profiling getters.  mcpc_run does not wait for the stream, with two
 * bounded exceptions: a run with MCPC_XOPT_ADAM uploads its bias-correction table from one of two pinned
 * staging buffers and waits (hipEventSynchronize) for the upload issued two Adam runs earlier if that
 * has still not executed; and the first run that accumulates Hebbian sums allocates the spill ring
 * (hipMalloc).  Device buffers that have to grow (per-step tables, energy partials; geometrically) are
 * replaced, the old ones retired behind an event and freed by a later run once that event has completed.
 * The library reads no environment variables.
 */
#ifndef MCPC_H
#define MCPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCPC_ABI_VERSION 4
#define MCPC_MAX_LATENT 6

/* status codes */
#define MCPC_OK 0
#define MCPC_EINVAL (-1)       /* bad argument / unsupported configuration */
#define MCPC_EHIP (-2)         /* a HIP runtime call failed */
#define MCPC_ENOMEM (-3)       /* configuration does not fit the device (LDS / HBM) */
#define MCPC_ESTATE (-4)       /* call order violated (e.g. run before bind) */

/* activation applied to x_l before the next Linear (reference utils/model.py:49-52) */
#define MCPC_ACT_IDENTITY 0
#define MCPC_ACT_RELU 1
#define MCPC_ACT_TANH 2

/* output loss (reference utils/model.py:17-33); masked variants = mask_start > 0 */
#define MCPC_LOSS_NONE 0       /* loss_fn=None or zero_fn */
#define MCPC_LOSS_GAUSSIAN 1   /* fe_fn: (1/var)*0.5*(out-y)^2, summed */
#define MCPC_LOSS_BERNOULLI 2  /* bernoulli_fn: BCEWithLogits, summed */

/* optimizer on x (reference pc_trainer.py:465-475,871-877) */
#define MCPC_XOPT_SGD 0        /* optim.SGD(lr), no momentum / weight decay */
#define MCPC_XOPT_ADAM 1       /* optim.Adam(lr, betas, eps), state reset per run */

/* noise source for the Langevin kick x += sqrt(noise_var*lr)*xi (reference utils/model.py:35-44) */
#define MCPC_NOISE_NONE 0      /* PC / MAP inference */
#define MCPC_NOISE_PHILOX 1    /* fused counter-based Philox4x32-10 + Box-Muller */
#define MCPC_NOISE_EXTERNAL 2  /* xi read from ext_noise[l] (parity tests, arbitrary generators) */

/* which steps write loss / layer energies (reference pc_trainer.py:776-797,835-836) */
#define MCPC_ENERGY_NONE 0
#define MCPC_ENERGY_LAST 1
#define MCPC_ENERGY_ALL 2

typedef struct mcpc_engine mcpc_engine;

/* Static description of the network and the shard of chains this engine owns. */
typedef struct mcpc_net_desc {
    int32_t abi_version;                 /* MCPC_ABI_VERSION */
    int32_t n_latent;                    /* number of PCLayers L, 1..MCPC_MAX_LATENT */
    int32_t n_in;                        /* width of the pseudo-input fed to the first Linear */
    int32_t sizes[MCPC_MAX_LATENT];      /* n_1..n_L, top latent first (reference utils/model.py:54-65) */
    int32_t acts[MCPC_MAX_LATENT];       /* MCPC_ACT_* applied to x_l */
    float ecoef[MCPC_MAX_LATENT];        /* c_l in energy c_l*0.5*(mu-x)^2 (reference pc_layer.py:17-18, figure_3.py:47-48) */
    int32_t n_out;                       /* width of the read-out Linear; 0 = model ends with a PCLayer */
    int32_t batch;                       /* chains held by this engine (local shard) */
    int32_t device;                      /* HIP device ordinal */
    int64_t spill_budget_bytes;          /* HBM budget for the Hebbian spill ring; 0 = default: room for 384 steps in three parts (17 GB at 6000
                                          * chains of cfg-M's net), at least 6 GiB, at most a quarter of the device's memory */
    const char* tuning;                  /* NULL, or developer overrides of the schedule heuristics as "key=value,key=value"
                                          * (parsed once by mcpc_create, not kept): ws=0|2|3 step kernel (0: the barrier kernel, the
                                          * fallback and the independent form parity checks replay the default against; 2: the in-place
                                          * wave-specialised kernel for every run; 3: the unified-wave kernel for the runs it serves -- fused
                                          * SGD / Adam updates -- or an error when its LDS plan does not fit; default: in-place, with the
                                          * unified-wave kernel for small networks and for zero-loss calls), u_row / u_gemm0 / u_kb / u_kbt /
                                          * u_eh / u_eb / u_ef=N (cost model the unified-wave kernel's rows are dealt by),
                                          * no_overlap=1, slot_cap=N, spill_gb=N, ring_parts=N,
                                          * flush_tail=N, flush_streams=1|2, cu_slack=N, dw_ksplit=N, ws_prio=0|1|2, stagger=N, no_lean=1,
                                          * no_ybits=1, overlay16=1, heb_fp32=1 (the Hebbian GEMM on the fp32 MFMA instead of its fp16
                                          * form), rr=0 (shards of more 16-chain units than CUs as ONE launch in hardware rounds instead of
                                          * the round schedule), rr_qmax=N (most steps per launch of the round schedule), no_xl=1 (state and
                                          * per-step constants of a workgroup's chains in global memory instead of LDS).  Unknown keys are
                                          * an error.  Used by A/B runs and by the tests that pin the kernel forms against each other. */
} mcpc_net_desc;

/* One train_on_batch call (or a slice of it).  Steps are numbered 0..T-1 inside the call. */
typedef struct mcpc_run_desc {
    int32_t T;                   /* total steps of the reference call (PCTrainer T), for 'last' semantics */
    int32_t t_begin;             /* first step executed by this run */
    int32_t n_steps;             /* steps executed by this run (t_begin + n_steps <= T) */

    /* Scalar hyper-parameters are DOUBLES (ABI 4): they are Python floats in the reference, and torch rounds each to fp32 exactly
     * once, where it is used (alpha = -lr of SGD's add_; 1 - beta1 of Adam's lerp_; -lr / (1 - beta1^t) of its addcdiv_; 1 / _var of
     * fe_fn).  The library rounds at the same places; an fp32 field would round before the subtraction / division instead
     * (1 - double(0.9f) != 1 - 0.9). */
    int32_t loss_kind;           /* MCPC_LOSS_* */
    int32_t mask_start;          /* first output column that contributes: n_out - round(n_out*perc); 0 = unmasked */
    double loss_var;             /* Gaussian variance (_var) */

    int32_t xopt_kind;           /* MCPC_XOPT_* */
    int32_t adam_step0;          /* Adam step count before this run (0 at the start of a call) */
    double lr;
    double beta1, beta2, eps;    /* Adam */

    int32_t update_x;            /* 1: fused x update (the fast path). 0: gradients only -> xgrad (generic callbacks) */

    int32_t noise_mode;          /* MCPC_NOISE_* */
    double noise_var;            /* random_step's `var` (2.0 = correct Langevin) */
    uint64_t seed;               /* Philox key */
    uint64_t step_base;          /* Philox step counter of step 0 of this call (advances across calls) */
    uint64_t chain_base;         /* global id of this shard's first chain (sharding-invariant noise) */
    const float* ext_noise[MCPC_MAX_LATENT]; /* MCPC_NOISE_EXTERNAL: [n_steps][batch][n_l] per layer */

    int32_t acc_begin, acc_end;  /* accumulate parameter-gradient sums over steps [acc_begin, acc_end) of the call */
    int32_t acc_reset;           /* 1: zero the sums when this run starts accumulating (reference pc_trainer.py:853-859) */

    int32_t energy_mode;         /* MCPC_ENERGY_* */
    double* energies_out;        /* [T or 1][MCPC_MAX_LATENT+2]: loss, E_1..E_L (unused = 0), overall; device pointer */

    int32_t rec_begin, rec_stride, rec_count; /* record x_t / outputs at t = rec_begin + k*rec_stride, k < rec_count */
    float* rec_x[MCPC_MAX_LATENT];            /* [rec_count][batch][n_l] or NULL (reference pc_trainer.py:440-445,772-774) */
    float* rec_out;                           /* [rec_count][batch][n_out] or NULL (is_return_outputs, :769-770) */

    float* xgrad[MCPC_MAX_LATENT];            /* update_x == 0: dF/dx_l -> [batch][n_l] */
} mcpc_run_desc;

/* lifetime -------------------------------------------------------------------------------- */
int mcpc_abi_version(void);
/* What this binary is (ABI 4; nothing in the reference to mirror): one line of space-separated key=value words,
 *   "libmcpc abi=4 arch=gfx950 csrc=<sha256[:16] of the kernel sources + this header> commit=<git short hash[+dirty]|unknown>
 *    exp=0|1 stamps=0|1 flags=[<compiler flags>]"
 * exp=1: the library was built with a timing-experiment switch (csrc/mcpc_build.h: it computes WRONG results on purpose and must
 * never be tested or benchmarked as the product; the Python binding refuses it, bench.py asserts exp=0).  stamps=1: the diagnostic
 * build with in-kernel phase stamps (correct results).  The string is static storage, valid for the life of the process. */
const char* mcpc_build_info(void);
const char* mcpc_last_error(void);
int mcpc_create(const mcpc_net_desc* desc, mcpc_engine** out);
int mcpc_destroy(mcpc_engine* e);

/* Parameters of Linear j (j = 0..L-1 predicts latent layer j+1; j = L is the read-out), torch
 * nn.Linear layout W[out][in] row-major, bias[out] or NULL.  The engine keeps the pointers
 * (borrowed storage) and re-packs them into MFMA fragment order at mcpc_params_changed().
 * Replaces: nn.Linear.forward inside self._model(self.inputs), pc_trainer.py:733. */
int mcpc_bind_params(mcpc_engine* e, int j, const float* W, const float* bias);
/* Re-pack all bound parameters (call after binding and after every optimizer_p.step()). */
int mcpc_params_changed(mcpc_engine* e, void* stream);

/* Pseudo-input [batch][n_in] (NULL = zeros, the reference's usual call) and target [batch][n_out].
 * Replaces: `inputs`, loss_fn_kwargs['_target'] of train_on_batch, pc_trainer.py:500-524. */
int mcpc_bind_inputs(mcpc_engine* e, const float* inputs, void* stream);
int mcpc_bind_target(mcpc_engine* e, const float* target, void* stream);

/* Latent state x_l, [batch][n_l] row-major (PCLayer._x, pc_layer.py:230,300).  load copies the
 * caller's tensors into the engine's padded state; store writes the current state back. */
int mcpc_load_state(mcpc_engine* e, const float* const* x, void* stream);
int mcpc_store_state(mcpc_engine* e, float* const* x, void* stream);

/* Adam moments of the x optimizer after a run with MCPC_XOPT_ADAM: exp_avg / exp_avg_sq per latent layer, [batch][n_l]
 * (torch.optim.Adam's per-parameter state, reference pc_trainer.py:465-475).  Lets the caller keep optimizer_x alive across
 * calls the way the reference does when neither reset flag of train_on_batch is set (pc_trainer.py:742-752). */
int mcpc_store_adam_state(mcpc_engine* e, float* const* m, float* const* v, void* stream);

/* The hot loop: n_steps iterations of pc_trainer.py:712-981 (+ random_step) on every chain. */
int mcpc_run(mcpc_engine* e, const mcpc_run_desc* run, void* stream);

/* Un-normalised parameter-gradient sums of Linear j accumulated by mcpc_run:
 *   dW[out][in] = scale * sum_t dF/dW(x_t),  db[out] = scale * sum_t dF/db(x_t)   (db may be NULL)
 * accumulate != 0 adds to the destination instead of overwriting (autograd's += into .grad).
 * Replaces: overall.backward()'s parameter part + the normalisation of pc_trainer.py:905-913. */
int mcpc_read_param_grads(mcpc_engine* e, int j, float* dW, float* db, float scale, int accumulate, void* stream);
/* Same, all Linears concatenated (W0,b0,W1,b1,...; absent biases skipped) into one flat buffer,
 * the bucket that is all-reduced once per call across shards. */
int mcpc_read_param_grads_flat(mcpc_engine* e, float* flat, int64_t n_floats, float scale, void* stream);
int64_t mcpc_param_count(const mcpc_engine* e);

/* ---- multi-GPU (the reference is single-device, `use_cuda = torch.cuda.is_available()`
 * figure_2.py:150): one process per GPU, the chains sharded, the weights replicated.  The ONLY collective of a learning
 * call is one sum of the gradient bucket over the shards, before the division by len(accumulate_p_at) * B_global
 * (pc_trainer.py:905-909 divides by len(inputs) = the whole batch).  For a host that is not on torch.distributed the
 * library drives RCCL itself (librccl.so.1 is loaded on first use; ncclAllReduce(sum, fp32) over xGMI):
 *   rank 0:   mcpc_comm_unique_id(id)         and ships the MCPC_COMM_ID_BYTES to the other ranks out of band
 *   all:      mcpc_comm_init(e, n_ranks, rank, id)            (collective: every rank must call it)
 *   per call: mcpc_read_param_grads_flat(e, flat, n, 1/(n_acc*B_global), stream); mcpc_allreduce_grads(e, flat, n, stream)
 * mcpc_allreduce_grads is asynchronous on `stream` and in place; with a communicator of one rank it is the identity.
 * mcpc_destroy releases the communicator. */
#define MCPC_COMM_ID_BYTES 128
int mcpc_comm_unique_id(void* id_out);
int mcpc_comm_init(mcpc_engine* e, int n_ranks, int rank, const void* id);
int mcpc_allreduce_grads(mcpc_engine* e, float* flat, int64_t n_floats, void* stream);
int mcpc_comm_destroy(mcpc_engine* e);

/* Fill out[batch][n_units] with the engine's Philox normals for (seed, step, layer): the device
 * generator exposed for bit-exactness tests against oracle/philox.py. raw != 0 writes the u32 stream. */
int mcpc_philox_normals(int device, uint64_t seed, uint64_t step, int layer, uint64_t chain_base,
                        int batch, int n_units, float* out, int raw, void* stream);

/* Synchronise `stream` and report device-side faults of the runs issued so far (the wave-specialised step kernel
 * bounds every intra-workgroup wait; a wait that runs out is recorded instead of hanging the GPU).
 * Returns MCPC_ESTATE if the last results must not be used.  The only host-synchronising call besides create/destroy. */
int mcpc_sync_check(mcpc_engine* e, void* stream);

/* Introspection for benchmarks / DESIGN.md: bytes of LDS per workgroup, chains per workgroup, workgroups of the shard (units
 * of `chains_per_wg` chains that hold at least one chain of the batch; a launch of the round schedule holds a part of them, see
 * mcpc_step_kernel_name), spill slots. */
int mcpc_query(const mcpc_engine* e, int32_t* lds_bytes, int32_t* chains_per_wg, int32_t* n_workgroups,
               int32_t* spill_slots);
/* Name of the step kernel this engine launches, as it appears in a rocprofv3 kernel trace (static string). */
const char* mcpc_step_kernel_name(const mcpc_engine* e);

/* Timing hooks.  While profiling is enabled (mcpc_set_profiling(e, 1); every call of it resets the tallies), mcpc_run
 * brackets every step-kernel launch with HIP events on its stream (a launch of the round schedule advances only the workgroups it
 * holds: it counts as steps x workgroups / all workgroups whole-shard steps).  The getter synchronises on the recorded events and
 * returns the summed time, the number of launches and the whole-shard steps they cover (rounded), accumulated over all runs since
 * profiling was enabled (at most 65 536 launches). */
int mcpc_set_profiling(mcpc_engine* e, int enable);
int mcpc_last_step_kernel_ms(mcpc_engine* e, float* ms, int32_t* n_launches, int64_t* n_steps);
/* The shader clock the chip held DURING the step-kernel launches bracketed since profiling was enabled: one wave of workgroup 0 of
 * every launch of the in-place kernel reads s_memtime (shader cycles) and s_memrealtime (100 MHz) at both ends of the launch; the
 * quotient of the sums is the clock under that load (MI355X lowers it under MFMA-dense work: 1.8-2.0 GHz against the 2.4 GHz peak).
 * 0 when no such launch has run.  Waits for the device. */
int mcpc_last_shader_clock_ghz(mcpc_engine* e, float* ghz);

/* Diagnostic (tests only; nothing of the product path calls it): fill the 160 KiB of LDS of EVERY compute unit of `device` with the
 * 32-bit pattern `word` (e.g. 0x7fa00000, a signalling NaN), then return once the fill has completed.  LDS is not cleared between
 * kernels on gfx950, so the next launch on each CU finds the pattern in whatever LDS it does not write itself.  Used by
 * tests/test_gpu_lds_poison.py to show that no result depends on LDS content the step kernels did not produce. */
int mcpc_debug_poison_lds(int device, uint32_t word, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MCPC_H */
