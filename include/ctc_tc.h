/*
 * ctc_tc.h - C ABI of the MI355X (gfx950) tensor-contraction backend, libctc_tc.so.
 *
 * Where this sits relative to the reference (zjing14/CTC): the engine has no
 * in-process operator table and no numeric API.  The only extension point it
 * offers is the process boundary - a plugin XML whose <command> text is
 * expanded (engine/src/ctc_plugin.cpp:166-206, engine/src/ctc_util.cpp:130-290)
 * and handed to system() (engine/src/ctc_engine.cpp:160,285).  The executable
 * `ctc_contract` that ships with this library is what such a plugin names; it is
 * a thin argv front-end over the entry points below.  The engine's own C++ class
 * headers (engine/include/ctc_engine.h:10-36, ctc_plugin.h:9-67) stay untouched.
 * Because the reference has no contraction entry point, none of the functions
 * below replaces a reference function one-for-one; each comment names the
 * reference call site whose role it takes over, or says that there is none.
 *
 * Conventions
 *   - plain C types only; every tensor pointer is a DEVICE pointer (HIP) unless
 *     the name says host; the caller owns every buffer;
 *   - return value 0 = success, negative = ctc_tc_status; the message of the
 *     last failure on the calling thread is ctc_tc_last_error();
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     execution entry points only enqueue work and return;
 *   - a plan is immutable after creation and may be used from several threads;
 *   - tensors are dense with explicit element strides; strides == NULL means
 *     packed with the LAST mode of the operand fastest.
 */
#ifndef CTC_TC_H
#define CTC_TC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTC_TC_MAX_MODES 16

typedef struct ctc_tc_plan ctc_tc_plan;

enum ctc_tc_dtype { CTC_TC_F64 = 0, CTC_TC_F32 = 1, CTC_TC_BF16 = 2 };

enum ctc_tc_status {
    CTC_TC_OK = 0,
    CTC_TC_EINVAL = -1,       /* bad argument */
    CTC_TC_ESPEC = -2,        /* malformed contraction spec */
    CTC_TC_EDEVICE = -3,      /* HIP runtime error or no gfx950 device */
    CTC_TC_ENOMEM = -4,
    CTC_TC_EUNSUPPORTED = -5, /* valid request this build cannot run */
    CTC_TC_EALIGN = -6        /* operand pointer not aligned as the plan requires */
};

/* mode classes reported by ctc_tc_plan_get(CTC_TC_Q_CLASSES) */
enum ctc_tc_mode_class {
    CTC_TC_CLASS_BATCH = 0,      /* in A, B and C */
    CTC_TC_CLASS_CONTRACTED = 1, /* in A and B only */
    CTC_TC_CLASS_FREE_M = 2,     /* in A and C only */
    CTC_TC_CLASS_FREE_N = 3      /* in B and C only */
};

/* integer outputs of the planner.  "mode ids" index the distinct letters of
 * the spec in order of first appearance (A, then B, then C). */
enum ctc_tc_query {
    CTC_TC_Q_NAMES = 0,        /* one letter (as int) per distinct mode */
    CTC_TC_Q_CLASSES = 1,      /* ctc_tc_mode_class per distinct mode */
    CTC_TC_Q_EXTENTS = 2,
    /* canonical GEMM view: batch/M/N in C order, K in A order (slowest first) */
    CTC_TC_Q_CANON_BATCH = 3,
    CTC_TC_Q_CANON_M = 4,
    CTC_TC_Q_CANON_N = 5,
    CTC_TC_Q_CANON_K = 6,
    /* canonical transposes (position p of the result takes position perm[p]):
     * A -> [batch,M,K], B -> [batch,K,N], [batch,M,N] -> C */
    CTC_TC_Q_PERM_A = 7,
    CTC_TC_Q_PERM_B = 8,
    CTC_TC_Q_PERM_C = 9,
    /* execution view chosen for the device (FASTEST mode first) */
    CTC_TC_Q_SWAPPED = 10,     /* 1: operands exchanged so that C's unit-stride mode is an N mode */
    CTC_TC_Q_EXEC_BATCH = 11,
    CTC_TC_Q_EXEC_M = 12,
    CTC_TC_Q_EXEC_N = 13,
    CTC_TC_Q_EXEC_K = 14,
    CTC_TC_Q_GEMM_DIMS = 15,   /* {M, N, K, batch} */
    CTC_TC_Q_TILE = 16,        /* {BM, BN, BK, threads, layoutA, layoutB, vecA, vecB, group, kernel, vecC, transposes:
                                  0 the permutation is fused into the operand loads, 1 ctc_tc_execute transposes A, B and
                                  C explicitly (a batch mode is the unit-stride mode of all three), 2 it copies A and B
                                  into padded K-major rows first (an operand would be read element by element under a
                                  large contraction) and writes C in place} */
    /* element-offset tables of the execution view: address of GEMM element
     * A(m,k) = A + ROW_A[m] + K_A[k] (+ BAT_A[b]), and so on */
    CTC_TC_Q_ROW_A = 17,
    CTC_TC_Q_K_A = 18,
    CTC_TC_Q_K_B = 19,
    CTC_TC_Q_COL_B = 20,
    CTC_TC_Q_ROW_C = 21,
    CTC_TC_Q_COL_C = 22,
    CTC_TC_Q_BAT_A = 23,
    CTC_TC_Q_BAT_B = 24,
    CTC_TC_Q_BAT_C = 25,
    CTC_TC_Q_FLOPS = 26,       /* {2*M*N*K*batch} */
    CTC_TC_Q_SCHEDULE = 27     /* needs a HIP device.  How a launch with 16-byte aligned operand bases runs on the
                                  current device: {tile rows, tile columns, tiles run whole, tiles of the last round
                                  that are cut along K, chunks per such tile, workspace bytes of that launch,
                                  kernel form (0: operands staged through registers; 1-4: staged by LDS-DMA with
                                  K-tiles of 64-byte rows / two images of 32-byte rows / one image of 32-byte rows /
                                  64-byte rows with ragged K-tiles), largest workspace bytes over the launch and its
                                  fall-backs to element loads for unaligned bases, stream-K shares (0: the cut tiles
                                  are cut into `chunks per such tile` equal chunks; > 0: the K-tile iterations of the
                                  cut tiles, taken as one sequence, are dealt out in that many equal shares),
                                  batch entries whose tiles the launch counts as ONE list (0: the batch is a grid
                                  dimension and the tile counts above are per entry; > 0: the counts of whole and cut
                                  tiles run over tile rows x tile columns x that many entries, so a batched launch's
                                  last round is balanced like a single product's; of the first launch when the batch
                                  takes several)} */
    ,
    CTC_TC_Q_SPANS = 28        /* {elements of A, of B, of C the plan addresses, counted from the pointer an execution
                                  is given: 1 + the largest element offset (the packed element count for packed
                                  operands; for a slab plan of ctc_tc_plan_shard, from the slab's first element)}.
                                  What a host that knows its buffers' sizes checks before it launches. */
};

/* operand layouts reported in CTC_TC_Q_TILE */
enum ctc_tc_layout {
    CTC_TC_LAYOUT_K_VEC = 0,     /* unit stride along the fastest K mode, 16-byte loads */
    CTC_TC_LAYOUT_OUTER_VEC = 1, /* unit stride along the fastest M (or N) mode, 16-byte loads */
    CTC_TC_LAYOUT_SCALAR = 2     /* no usable unit stride: element loads */
};

/* kernel families reported in CTC_TC_Q_TILE */
enum ctc_tc_kernel {
    CTC_TC_KERNEL_MFMA_GEMM = 0,        /* LDS-tiled MFMA GEMM with permute-on-load */
    CTC_TC_KERNEL_SKINNY_SAME = 1,      /* M <= 16 stream, B and C contiguous along the same N mode */
    CTC_TC_KERNEL_SKINNY_TRANSPOSE = 2  /* M <= 16 stream, B and C contiguous along different N modes */
};

/* Library version: major*10000 + minor*100 + patch.  No reference counterpart. */
int ctc_tc_version(void);

/* Message of the last failure on this thread ("" if none).  Takes the role of the
 * engine's `throw "literal"` + top-level print (engine/src/execute_engine.cpp:96-99). */
const char *ctc_tc_last_error(void);

/*
 * Process-wide kernel-selection knobs, for measurements and tests (production code never needs them):
 *   "stream_mode"   0 (default): contractions with a short K under a large C run on the persistent
 *                   deferred-store kernel, everything else on the one-tile-per-workgroup kernel;
 *                   1: never the persistent kernel; 2: the persistent kernel whenever it can run the
 *                   launch (f64 / f32, 16-byte-vector layouts, 16-byte stores of C, beta == 0, one
 *                   batch entry, K > 32);
 *   "stream_max_k"  largest K the automatic choice sends to the persistent kernel (default 128; fp32: at most 96);
 *   "stream_f64_dma" 1 (default): fp64 launches that qualify (even M and N, one stride per operand along
 *                   K) take the persistent kernel's LDS-DMA form; 0: its register-staged form;
 *   "dma_gemm"      1 (default): fp64 / fp32 launches on the 128 x 128 tile whose operands both have a
 *                   16-byte-vector layout and whose k offsets are affine inside every aligned group of 4
 *                   (fp64) or 8 (fp32) contracted indices, K a multiple of that, stage their operands by
 *                   LDS-DMA; 0: always through registers;
 *   "stream_k"      how the LDS-DMA kernel's schedule treats a tile count that is no multiple of the chip's
 *                   workgroup slots: 0 (default) by estimated cost, either the tiles of the last round cut
 *                   into equal K chunks or the K-tile iterations of the last one to two rounds dealt out in
 *                   equal shares (stream-K); 1: never shares; 2: shares whenever the launch allows;
 *   "stream_k_group" consecutive tiles whose shares walk K together (1 .. 64); 0 (default): 16, or 8 where that
 *                   leaves less padding;
 *   "small_batched" 1 (default): a batch of at least 512 products with M, N <= 32 and K <= 64 (M, N <= 16: K <= 4096), operands with one unit stride each (fp32: not
 *                   32 x 32 under a K above 16) runs on the grouped kernel, one wave per product; 0: on the tiled kernels;
 *   "pad_transposes" 1 (default): a plan whose CTC_TC_Q_TILE reports transposes = 2 runs through the padded copies;
 *                   0: it runs the fused kernel with element loads (A/B runs).
 * No reference counterpart.
 */
int ctc_tc_set_option(const char *name, int64_t value);
/* Value of a knob above, or of a read-only counter of tiled-GEMM launches since the library was loaded:
 * "launches_classic" (register-staged), "launches_dma_gemm", "launches_stream" (persistent,
 * register-staged), "launches_stream_f64_dma", "launches_bf16_ring" (the bf16 256-tile ring kernel, a
 * subset of "launches_classic"), "padded_executions" (executions that went through the padded copies; their GEMM
 * counts in the launch counters as well), "launches_grouped" (batched plans that took the grouped kernel).
 * Negative ctc_tc_status for an unknown name. */
int64_t ctc_tc_get_option(const char *name);

/* Number of usable HIP devices (0 when there is none; never fails). */
int ctc_tc_device_count(void);

/*
 * Contraction planner (host only, needs no GPU).  spec is "aeim,ebmj->abij";
 * mode_names/extents are parallel arrays of n_ext entries giving the extent of
 * every letter.  No reference counterpart (the reference has no planner); it is
 * reached from the plugin command line the engine builds in
 * engine/src/ctc_plugin.cpp:175-206.
 */
int ctc_tc_plan_create(const char *spec, int n_ext, const char *mode_names, const int64_t *extents,
                       const int64_t *strides_a, const int64_t *strides_b, const int64_t *strides_c,
                       int dtype, ctc_tc_plan **plan);

int ctc_tc_plan_destroy(ctc_tc_plan *plan);

/* Human-readable plan (classes, orders, tile, kernel variant), NUL-terminated.
 * Returns the length that would have been written.  Mirrors the role of
 * Plugin::print / Workflow::print (engine/src/ctc_plugin.cpp:261-278). */
int ctc_tc_plan_describe(const ctc_tc_plan *plan, char *buf, size_t buflen);

/* Integer planner outputs.  Writes at most cap values, returns how many exist
 * (negative on error). */
int64_t ctc_tc_plan_get(const ctc_tc_plan *plan, int query, int64_t *out, int64_t cap);

/* The plan whose tiled GEMM ctc_tc_execute launches for `plan`: `plan` itself, or the sub-plan it owns when it runs
 * through explicit transposes or padded copies (CTC_TC_Q_TILE's transposes 1 / 2; built here on first use, host
 * only).  CTC_TC_Q_SCHEDULE of `plan` describes the fused kernel (what ctc_tc_execute_noalloc launches);
 * CTC_TC_Q_SCHEDULE of the returned plan describes the GEMM behind the copies.  Owned by `plan`: do not destroy.
 * NULL on failure.  No reference counterpart. */
ctc_tc_plan *ctc_tc_plan_gemm_plan(ctc_tc_plan *plan);

/*
 * Plan for one shard of a contraction split along `mode` into nranks equal
 * contiguous slabs (extent must divide evenly).  Element offsets of the slab
 * inside the full A, B, C are returned in off_a/off_b/off_c (0 for an operand
 * that does not hold the mode).  A free mode needs no exchange afterwards; a
 * contracted mode leaves every rank with a full-size partial C to be summed
 * (reduce-scatter / all-reduce by the caller).  No reference counterpart: the
 * reference is single-device (apps/fastr/IndelRealigner.cpp:805-829).
 */
int ctc_tc_plan_shard(const ctc_tc_plan *plan, char mode, int rank, int nranks,
                      ctc_tc_plan **shard, int64_t *off_a, int64_t *off_b, int64_t *off_c);

/*
 * ---- Contracted-mode sharding with RCCL (one rank per GPU) ---------------------------------------
 * No reference counterpart: the reference is single-device (apps/fastr/IndelRealigner.cpp:805-829).
 *
 * Every rank holds its slab of A and of B along a CONTRACTED mode (packed, extent/p along that mode)
 * and computes a partial C; the partials are summed over the ranks.  C is cut into chunks along its
 * slowest mode and the sum of chunk i runs on a second stream while chunk i+1 is multiplied.
 *   CTC_TC_REDUCE_SCATTER  rank r ends up with the flat range [r |C|/p, (r+1) |C|/p) of the packed C in
 *                          its output buffer (|C|/p elements); needs a workspace of |C| elements for
 *                          the partial result.  Chunks are interleaved so that this ownership holds
 *                          (needs the extent of C's slowest mode to be a multiple of p; otherwise one
 *                          un-pipelined reduce-scatter of the whole partial result is used).
 *   CTC_TC_ALL_REDUCE      every rank ends up with the whole C in its output buffer (|C| elements);
 *                          no workspace.
 * Communicators are plain RCCL communicators (ncclComm_t) passed as void*.  A host that has none
 * creates them here: rank 0 calls ctc_tc_comm_unique_id and sends the CTC_TC_UNIQUE_ID_BYTES bytes to
 * the other ranks by whatever means it has; every rank then calls ctc_tc_comm_init_rank with its
 * device current.  A single process that drives several GPUs calls ctc_tc_comm_init_all.
 */
#define CTC_TC_UNIQUE_ID_BYTES 128
enum ctc_tc_reduce { CTC_TC_REDUCE_SCATTER = 0, CTC_TC_ALL_REDUCE = 1 };
enum ctc_tc_sharded_query {
    CTC_TC_SQ_CHUNKS = 0,           /* number of chunks the execution is cut into */
    CTC_TC_SQ_C_ELEMS = 1,          /* |C| */
    CTC_TC_SQ_OUT_ELEMS = 2,        /* elements of the rank's output buffer */
    CTC_TC_SQ_WORKSPACE_ELEMS = 3,  /* elements of the workspace (0: none needed) */
    /* per chunk (for hosts that emulate or inspect the pipeline): element offsets of the chunk's GEMM
     * into the rank-local A and B, into the partial buffer (reduce-scatter) or C (all-reduce), into
     * the rank's output (reduce-scatter), and the elements of C the chunk covers over all ranks */
    CTC_TC_SQ_CHUNK_OFF_A = 16,
    CTC_TC_SQ_CHUNK_OFF_B = 17,
    CTC_TC_SQ_CHUNK_OFF_PARTIAL = 18,
    CTC_TC_SQ_CHUNK_OFF_OUT = 19,
    CTC_TC_SQ_CHUNK_ELEMS = 20
};
typedef struct ctc_tc_sharded ctc_tc_sharded;

int ctc_tc_comm_unique_id(void *id128);
int ctc_tc_comm_init_rank(void **comm, int nranks, int rank, const void *id128);
int ctc_tc_comm_init_all(void **comms, int ndev, const int *devices /* NULL: 0..ndev-1 */);
int ctc_tc_comm_destroy(void *comm);
/* Ends a communicator whose collectives can no longer complete (a peer rank failed before entering them):
 * collectives in flight on it return with an error instead of waiting for ever, and the communicator is
 * freed (do not destroy it afterwards).  ncclCommAbort. */
int ctc_tc_comm_abort(void *comm);

/* extents are those of the WHOLE contraction; `mode` must be a contracted mode whose extent is a
 * multiple of nranks; nchunks is the wanted pipeline depth (the largest count <= nchunks that divides
 * the rows evenly is taken; 1 = no overlap).  Host only, needs no GPU. */
int ctc_tc_sharded_create(const char *spec, int n_ext, const char *mode_names, const int64_t *extents,
                          int dtype, char mode, int rank, int nranks, int nchunks, int reduce,
                          ctc_tc_sharded **out);
int ctc_tc_sharded_destroy(ctc_tc_sharded *s);
int64_t ctc_tc_sharded_get(const ctc_tc_sharded *s, int query, int64_t chunk);
/* the plan of one chunk's GEMM (owned by `s`); its C is the chunk's piece of the partial buffer */
ctc_tc_plan *ctc_tc_sharded_chunk_plan(const ctc_tc_sharded *s, int chunk);
/* uploads every chunk's tables and creates the communication stream on the current device */
int ctc_tc_sharded_prepare(ctc_tc_sharded *s);
/* C_out = alpha * sum over ranks of contract(A_local, B_local), enqueued: the GEMMs on `stream`, the
 * collectives on the object's own stream; `stream` waits for the last collective before the call's
 * successors run.  nccl_comm may be NULL when nranks == 1, and for the all-reduce form (the call
 * then only multiplies and leaves this rank's partial C in C_out for the caller to sum).  beta is not offered: a C that every rank
 * adds in would be counted p times. */
int ctc_tc_sharded_execute(ctc_tc_sharded *s, void *nccl_comm, double alpha, const void *A_local,
                           const void *B_local, void *C_out, void *workspace, int64_t workspace_bytes,
                           void *stream);

/* Set-up of a plan on the current device, ahead of the first execution: uploads the offset
 * tables, allocates the workspace the plan's executions will use (split-K partial sums for every
 * layout variant an execution may fall back to; for a plan that runs through explicit transposes
 * also the packed sub-plan, the permutation tables and the three packed tensors) and resolves the
 * address of this device's fixed scratch buffer (see ctc_tc_execute_noalloc).  After it,
 * ctc_tc_execute on this device allocates nothing and never waits for the device.  Optional for
 * ctc_tc_execute (which does the same on first use), required before ctc_tc_execute_noalloc.
 * No reference counterpart; the nearest thing is the once-per-run OpenCL set-up in
 * apps/fastr/IndelRealigner.cpp:1219-1232. */
int ctc_tc_plan_prepare(ctc_tc_plan *plan);

/*
 * C = alpha * contract(A, B) + beta * C on the current device, enqueued on
 * `stream`.  A, B, C are device pointers to the plan's dtype.  Takes the role of
 * the device launch in apps/fastr/IndelRealigner.cpp:805-829 (clean_gpu).
 */
int ctc_tc_execute(ctc_tc_plan *plan, double alpha, const void *A, const void *B,
                   double beta, void *C, void *stream);

/*
 * ctc_tc_execute for callers that cannot allocate device memory: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  It has NO counterpart in the reference.
 * Same arguments and result as ctc_tc_execute, with these differences:
 *   - it calls no hipMalloc / hipFree and no allocating library routine; the only device memory
 *     beyond A, B and C it touches is a fixed scratch buffer of ctc_tc_noalloc_scratch_bytes()
 *     bytes (64 MiB) that is part of the library's code object, one copy per GPU, held until the
 *     process ends; the split-K partial sums of the last, partly filled round of tiles go there.
 *     This is what loading libctc_tc.so costs: every process that runs a kernel of the library on a
 *     GPU holds 64 MiB of that GPU's memory (0.02 % of an MI355X's 288 GB) from the first launch on,
 *     whether it ever calls this entry point or not.  64 MiB is one 128 x 128 fp64 slab per
 *     workgroup slot of the chip: what a stream-K schedule of the LDS-DMA GEMM kernel asks for at most
 *     (schedules that cut the last round of tiles into equal K chunks can ask for more);
 *   - ctc_tc_plan_prepare must have been called for this plan on the current device (it uploads
 *     the tables and resolves the scratch buffer's device address); otherwise CTC_TC_EINVAL;
 *   - a contraction whose schedule needs more scratch than that is rejected with
 *     CTC_TC_EUNSUPPORTED (ctc_tc_plan_get(CTC_TC_Q_SCHEDULE) reports the bytes a plan needs);
 *   - every call on one GPU, from any plan, stream or thread, shares that GPU's buffer: calls that
 *     overlap in time on one GPU race.  Order them (one stream, or events) yourself.  Nothing in
 *     the buffer has to be cleared between calls: every partial sum is written before it is read;
 *   - a plan that ctc_tc_execute would run through explicit transposes runs the fused kernel here.
 */
int64_t ctc_tc_noalloc_scratch_bytes(void);
int ctc_tc_execute_noalloc(ctc_tc_plan *plan, double alpha, const void *A, const void *B,
                           double beta, void *C, void *stream);

/*
 * Mode permutation on the device: out mode p = in mode perm[p]; both packed,
 * last mode fastest; elem_size 2, 4 or 8 bytes.  No reference counterpart.
 */
int ctc_tc_permute(const void *in, void *out, int rank, const int64_t *extents_in,
                   const int *perm, int elem_size, void *stream);

/*
 * Explicit transpose-transpose-GEMM-transpose execution of the same plan
 * through caller-provided workspace (bytes >= ctc_tc_ttgt_workspace_bytes):
 * permutes A and B to [batch,M,K]/[batch,K,N], runs the GEMM on packed
 * operands and permutes the result into C.  Exists to measure what the fused
 * permute-on-load path saves.  No reference counterpart.
 */
int64_t ctc_tc_ttgt_workspace_bytes(const ctc_tc_plan *plan);
int ctc_tc_execute_ttgt(ctc_tc_plan *plan, double alpha, const void *A, const void *B,
                        double beta, void *C, void *workspace, int64_t workspace_bytes,
                        void *stream);

/*
 * Grouped small contractions (ab = ak,kb per problem) in ONE launch: problem p computes
 * C_p[m x n] = alpha * A_p[m x k] * B_p[k x n] + beta * C_p, row-major with leading
 * dimensions, at element offsets off_a/off_b/off_c from the common bases A, B, C.
 * `problems` is a DEVICE array of `count` descriptors.  Extents are arbitrary (meant to be
 * a few tens).  The nearest thing in the reference is the way FASTR batches >= 32 regions
 * into one device launch (apps/fastr/IndelRealigner.cpp:93-94,1305-1313,805-829).
 */
typedef struct ctc_tc_small_gemm {
    int32_t m, n, k;
    int32_t lda, ldb, ldc;
    int64_t off_a, off_b, off_c;
} ctc_tc_small_gemm;

int ctc_tc_grouped_contract(const ctc_tc_small_gemm *problems, int count, int dtype, double alpha,
                            const void *A, const void *B, double beta, void *C, void *stream);

/*
 * Synthetic operand generator on the device: element k (packed order) =
 * uniform[-1,1) from splitmix64(seed + first + k), bit-identical to the CPU
 * oracle's generator.  No reference counterpart.
 */
int ctc_tc_fill(void *dev, int64_t n, uint64_t seed, int64_t first, int dtype, void *stream);

/*
 * Average device time in milliseconds of `iters` back-to-back executions of the
 * plan on `stream`, measured with HIP events recorded on that stream (after
 * `warmup` untimed executions).  No reference counterpart; the reference's
 * nearest analogue is its gettimeofday phase timers
 * (apps/fastr/IndelRealigner.cpp:74-91).
 */
int ctc_tc_time_execute(ctc_tc_plan *plan, double alpha, const void *A, const void *B,
                        double beta, void *C, void *stream, int warmup, int iters,
                        double *avg_ms);
/* The same with one event between every two executions: ms_each[i] = device time of execution i
 * (iters <= 4096 values), so that a caller can report a median beside the average.  No reference
 * counterpart. */
int ctc_tc_time_execute_each(ctc_tc_plan *plan, double alpha, const void *A, const void *B,
                             double beta, void *C, void *stream, int iters, double *ms_each);

#ifdef __cplusplus
}
#endif
#endif /* CTC_TC_H */
