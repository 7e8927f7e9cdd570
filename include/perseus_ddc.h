/*
 * perseus_ddc.h -- thin C ABI over the MI355X (gfx950) I/Q ingest + decimation
 * kernels.  Plain pointers and sizes only; no C++/torch types.
 *
 * What each entry point stands in for in the reference (libperseus-sdr):
 *
 *   pddc_unpack24_f32      examples/perseustest.c:466-502  user_data_callback_c_f
 *   pddc_unpack24_i32      examples/perseustest.c:432-460  user_data_callback_c_u
 *                          (dup. examples/simple.c:33-61)
 *   pddc_pack24_f32        (none) inverse of the above: what the FPGA emits on USB endpoint
 *                          0x82 -- 24-bit packed samples at the selected rate -- so that an
 *                          unmodified client sees the wire format it expects (perseustest.c:434)
 *   pddc_nco_freg          perseus-sdr.c:584   tuning word written to the FPGA
 *   pddc_pipeline_*        the FPGA DDC itself (NCO mix + decimating FIR chain)
 *                          that perseus_set_sampling_rate() selects by bitstream
 *                          (perseus-sdr.c:776-867) and perseus_set_ddc_center_freq()
 *                          tunes (perseus-sdr.c:556-619); no software model of it
 *                          exists in the reference, so its arithmetic is defined
 *                          by oracle/perseus_oracle.c.
 *   pddc_pipeline_push_host   what perseus-in.c:206-207 hands to the client
 *                          callback, batched (see perseus-sdr.h in this directory
 *                          for the drop-in callback API built on top of this).
 *   pddc_pipeline_push_host_async / _wait_ticket / pddc_host_alloc
 *                          the ring of 8 in-flight USB transfers of perseus-in.c:39-118
 *                          (queue create / submit / resubmit), as two pinned batches in
 *                          flight: copy in, kernels and copy out of neighbouring batches overlap
 *   pddc_pipeline_seek     (none) positions a pipeline inside ONE stream so that several
 *                          GPUs can take contiguous time chunks of it
 *
 * Sample formats
 *   packed : 6 bytes / complex sample, I0 I1 I2 Q0 Q1 Q2, 24-bit two's
 *            complement little endian (examples/perseustest.c:449-455)
 *   f32    : interleaved float32 I,Q  (8 bytes / sample), range [-1.00000012, 1]
 *   i32    : interleaved int32 I,Q, MSB aligned (multiples of 256)
 *
 * All functions return PDDC_OK (0) or a negative PDDC_E* code, and set a
 * thread-local message readable through pddc_last_error().  There is NO CPU
 * fallback: without a usable HIP device every compute entry point fails with
 * PDDC_ENODEV.
 *
 * "d_" pointers are device (HBM) addresses, "h_" pointers host addresses.
 * `stream` is a hipStream_t passed as void* (NULL = the null stream).
 */
#ifndef PERSEUS_DDC_H
#define PERSEUS_DDC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libperseus_ddc.so is built with -fvisibility=hidden: what this header declares is ALL the library exports (its C++
 * internals -- kernel stubs, launchers, the pipeline classes -- stay out of the dynamic symbol table) */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define PDDC_OK          0
#define PDDC_EINVAL     -1   /* bad argument (size, alignment, NULL)          */
#define PDDC_ENODEV     -2   /* no HIP device / device index out of range      */
#define PDDC_EHIP       -3   /* a HIP runtime call failed                      */
#define PDDC_ENOMEM     -4   /* allocation failed                              */
#define PDDC_ECAPACITY  -5   /* output buffer too small                        */
#define PDDC_ESTATE     -6   /* call not valid in the pipeline's current state */
#define PDDC_ECOMM      -7   /* an RCCL call failed                             */

#define PDDC_ADC_CLK_HZ        80000000.0   /* perseus-sdr.h:44 */
#define PDDC_MAX_STAGES        4
#define PDDC_MAX_TAPS          4096         /* rational (interp > 1) stages       */
#define PDDC_MAX_TAPS_DECIM    1024         /* plain decimators                   */
#define PDDC_FAST_MAX_TAPS     256          /* fused decimate-by-8 kernel      */
#define PDDC_PACKED_BYTES      6
#define PDDC_INPUT_GRANULE     8            /* process(): nsamples % 8 == 0    */

/* pipeline flags */
#define PDDC_F_MIX         0x1u   /* NCO complex mix before stage 0            */
#define PDDC_F_TAPS_FP16   0x2u   /* binary16 taps (config 5): rounded to binary16; a /8 first stage without NCO on the
                                     matrix cores (k_fir_i8x's plain form) then holds them on the device AS binary16, 2
                                     bytes a tap, and its matrix waves quantise them themselves; the other kernels keep
                                     the rounded values in fp32, tuned tables are built on the host from them */
#define PDDC_F_NO_FAST     0x4u   /* force the generic kernels (testing)       */
#define PDDC_F_OUT_PACKED24 0x8u  /* process()/push_host() emit 24-bit packed (6 B/sample)
                                     instead of float32: "FPGA emulation"      */

typedef struct pddc_pipeline pddc_pipeline;

typedef struct {
    int          decim;    /* decimation factor D (M of a rational L/M stage), >= 1 */
    int          ntaps;    /* 1 .. PDDC_MAX_TAPS_DECIM (PDDC_MAX_TAPS if interp > 1) */
    const float *taps;     /* h[0..ntaps-1], host memory, copied               */
    int          interp;   /* 0 or 1: plain decimator.  L > 1: rational resampler
                              y[m] = sum_j h[j*L + (m*D mod L)] * x[floor(m*D/L) - j]
                              (upsample by L, filter, keep every D-th); used for the
                              reference's non-integer rates  */
} pddc_stage_desc;

/* ---- library ------------------------------------------------------------ */
int         pddc_version(void);
const char *pddc_last_error(void);
int         pddc_device_count(void);                 /* >= 0, or PDDC_E*      */

/* ---- tuning word (perseus-sdr.c:584) ------------------------------------ */
uint32_t    pddc_nco_freg(double center_freq_hz, double adc_clk_hz);

/* ---- stateless kernels on device-resident buffers ------------------------ */
/* d_packed must be 16-byte aligned; nsamples may be any value >= 0.          */
int pddc_unpack24_f32(const void *d_packed, size_t nsamples, void *d_out_f32, void *stream);
int pddc_unpack24_i32(const void *d_packed, size_t nsamples, void *d_out_i32, void *stream);
/* float32 I/Q -> 24-bit packed: code = clamp(rint(x*8388607), -2^23, 2^23-1).  */
int pddc_pack24_f32(const void *d_in_f32, size_t nsamples, void *d_out_packed, void *stream);
/* synthetic source of BASELINE.md section 3: byte k of the LCG stream
 * s=s*1664525+1013904223, byte=s>>24, starting `byte_offset` bytes in.        */
int pddc_synth_lcg(void *d_dst, size_t nbytes, uint32_t seed, uint64_t byte_offset, void *stream);

/* ---- device memory helpers for C hosts (tests/bench use torch instead) ----
 * No reference counterpart: the reference's only buffers are the eight libusb transfer buffers of a queue in host
 * memory (perseus-in.c:67-110, input_queue_create); everything here exists because the DSP moved from the FPGA
 * into HBM. */
int pddc_set_device(int device);
int pddc_malloc(void **d_ptr, size_t nbytes);
int pddc_free(void *d_ptr);
/* Device memory for a buffer that is streamed AGAINST d_partner (one read while the other is written -- the packed
 * input and the float output of a pipeline): HBM is laid out in a few classes of large extents, and such a pair runs
 * ~8 % faster (k_fir8 127 taps: 0.340 instead of 0.367 ms) when the two buffers lie in extents of different classes;
 * allocations made one after the other usually share one.  This allocates candidates 8 GiB apart (spacers, freed
 * again), times a read+write probe stream between the partner and each, stops when both speeds have been seen or after
 * max_candidates, and returns the fastest (free it with pddc_free).  *ms_best / *ms_worst: the probe's time per launch
 * for the kept and for the slowest candidate.  Buffers under 1 MiB, a partner under 64 MiB, or max_candidates <= 1: a
 * plain allocation.  A buffer between 1 MiB and 1 GiB gets a 1 GiB allocation (the probe must write past the 256 MB
 * last-level cache to see the HBM), because small write streams matter too: the fused first two stages of the /320
 * cascade write 1/48 of what they read and still run at 0.292 or 0.330 ms depending on where that small buffer lies.
 * Candidates and spacers never take more than half of the free memory; a probe that fails leaves a plain allocation.
 * pddc_pipeline_place_buffers does this for a pipeline's own inter-stage buffers (process() never searches).         */
int pddc_malloc_apart(void **d_ptr, size_t nbytes, const void *d_partner, size_t partner_bytes, int max_candidates,
                      float *ms_best, float *ms_worst);
/* Where in ONE large allocation of the caller's (tens of GiB; 288 GB of HBM make it affordable) does a pipeline's write
 * side go?  The arena is cut into slots of slot_bytes; the packed batch lies at the arena's start (fill it BEFORE the
 * call), the write side at out_offset of a slot.  The probe is the pipeline's OWN first kernel: one fused stage writes its
 * nsamples / D outputs there; a cascade's inter-stage workspace (pddc_pipeline_workspace_size bytes) is set there with
 * pddc_pipeline_set_workspace and stays at the chosen slot -- the caller puts the output behind it.  Probed: the slot
 * right behind the input ("first come"), +32 / +48 / +64 GiB, every slot only if none of those gains 3 %; 0.1-0.2 s.
 * *ms_first_come / *ms_best: the kernel's time at slot 1 and at the returned slot.  Stream state is not advanced.
 * Worth 1-8 % (which slots are fast is a property of the process's physical layout, found, not modelled); a host that
 * does not care skips it.  This is the one placement entry point for caller-owned memory: rounds 2-4 also had
 * pddc_arena_search / pddc_arena_place, which ranked the slots with a read+write MODEL stream -- on one box in a dozen it
 * ranked them opposite to the kernel it stood for, so they are gone.  (No reference counterpart.)                  */
int pddc_pipeline_arena_place(pddc_pipeline *p, void *d_arena, size_t arena_bytes, size_t slot_bytes, size_t nsamples,
                              size_t out_offset, size_t *out_slot, float *ms_first_come, float *ms_best, int *nprobes,
                              void *stream);
int pddc_memcpy_h2d(void *d_dst, const void *h_src, size_t nbytes, void *stream);
int pddc_memcpy_d2h(void *h_dst, const void *d_src, size_t nbytes, void *stream);
int pddc_stream_sync(void *stream);

/* ---- DDC pipeline: [NCO mix] -> stage 0 -> ... -> stage n-1 --------------- */
int pddc_pipeline_create(pddc_pipeline **out, int device,
                         const pddc_stage_desc *stages, int nstages, uint32_t flags);
int pddc_pipeline_destroy(pddc_pipeline *p);
/* zero FIR histories, decimation phases and the NCO sample counter            */
int pddc_pipeline_reset(pddc_pipeline *p);

/* Inter-stage buffers of a cascade from CALLER-provided device memory instead of the pipeline's own allocations, so that
 * the host decides where they lie (the buffer a stage writes should sit in another HBM extent class than the batch the
 * stage reads -- see pddc_malloc_apart; bench.py cuts input, workspace and output from one arena and keeps the fastest
 * arrangement).  d_ws: 256-byte aligned, nbytes >= pddc_pipeline_workspace_size(p, max_nsamples); batches of more than
 * max_nsamples are then refused (PDDC_ECAPACITY).  Synchronises the device; call between batches.  d_ws == NULL
 * returns to own allocations.  The memory stays the caller's; stream state (histories, phases) is not touched.
 * (No reference counterpart: libperseus-sdr's buffers are libusb transfers, perseus-in.c:67-110.)               */
size_t pddc_pipeline_workspace_size(const pddc_pipeline *p, size_t max_nsamples);
int pddc_pipeline_set_workspace(pddc_pipeline *p, void *d_ws, size_t nbytes, size_t max_nsamples);
/* The same decision for the pipeline's OWN inter-stage buffers (a host that does not manage a workspace): allocates
 * them now, for batches of up to max_nsamples samples read from d_packed, each through the candidate walk of
 * pddc_malloc_apart against the buffer its writer reads.  Probes run on `stream`; candidates and spacers take at most
 * half of the free device memory and are freed again; a failing probe leaves a plain allocation.  About a second per
 * buffer, once.  pddc_pipeline_process itself never searches: without this call (or a workspace) the buffers are plain
 * allocations made when the first batch arrives.  (No reference counterpart.)                                    */
int pddc_pipeline_place_buffers(pddc_pipeline *p, const void *d_packed, size_t max_nsamples, void *stream);
/* reset, then place the stream at absolute input sample `abs_sample` with zero history:
 * the NCO phase and every stage's decimation phase are those of a stream that started at
 * sample 0.  This is what lets ONE stream be cut into time chunks for several GPUs
 * (SURVEY.md 8e (2)): a rank seeks to (chunk start - halo), processes halo + chunk and drops
 * the first halo/decimation outputs; the NCO needs no hand-over because its phase is a pure
 * function of the absolute index (perseus-sdr.c:584).  abs_sample must lie on an output
 * boundary of every stage (a multiple of the product of the decimation factors is enough)
 * and be a multiple of PDDC_INPUT_GRANULE.                                                */
int pddc_pipeline_seek(pddc_pipeline *p, uint64_t abs_sample);
/* New tuning word from the next sample on.  The NCO is a phase accumulator, like the FPGA's:
 * phase(n) = n*freg + offset (mod 2^32), and a retune at sample n moves the offset by
 * n*(freg_old - freg_new) so that the phase is continuous there (no phase jump on retune).
 * reset()/seek() clear the offset (phase of a stream that began at sample 0 with this word). */
int pddc_pipeline_set_freg(pddc_pipeline *p, uint32_t freg);
uint32_t pddc_pipeline_get_phase_offset(const pddc_pipeline *p);
int pddc_pipeline_set_center_freq(pddc_pipeline *p, double center_freq_hz);
int pddc_pipeline_set_taps(pddc_pipeline *p, int stage, const float *taps, int ntaps);
uint32_t pddc_pipeline_get_freg(const pddc_pipeline *p);
int pddc_pipeline_total_decim(const pddc_pipeline *p);
/* upper bound of outputs a process() of nsamples_in can produce               */
size_t pddc_pipeline_max_output(const pddc_pipeline *p, size_t nsamples_in);
/* exactly what the NEXT process()/push of nsamples_in will produce (depends on the
 * decimation phases the stream is at)                                          */
size_t pddc_pipeline_next_output(const pddc_pipeline *p, size_t nsamples_in);
/* 1 if stage 0 runs the fused unpack+mix+polyphase kernel for this geometry   */
int pddc_pipeline_uses_fused(const pddc_pipeline *p);
/* nonzero if a batch of nsamples_in would run stage 0 on the int8 matrix cores (the wire bytes are the operand, the
 * taps are quantised to 2^-31 of the largest one; same history, same outputs to 1e-7 of full scale)               */
int pddc_pipeline_stage0_on_i8(const pddc_pipeline *p, size_t nsamples_in);
/* The return value says WHICH kernel: 0 none (k_fir8), 2 k_fir_i8x (1 was round 3's k_fir_i8, retired in round 5: one
 * kernel family) -- without NCO its plain form, one tap table; with it the NCO folded into the taps, y[m] = LO(n0 + 8m) sum_k (h[k] e^{+j theta k}) x_raw[8m - k]: complex taps on the raw
 * integer planes, one float rotation per output -- which every tuned (PDDC_F_MIX) decimate-by-8 first stage of 1..256
 * taps runs on, i.e. every pipeline the drop-in API builds behind perseus_set_ddc_center_freq (perseus-sdr.c:556-619);
 * with a decimate-by-8 second stage of <= 64 taps behind a first stage of <= 128 and whole 8192-sample tiles, both are one kernel
 * (pddc_pipeline_uses_fused_pair answers 2).  The one batch whose history window straddles a retune takes k_fir8.   */
/* The operands k_fir_i8x reads, as the library builds them (host arithmetic, no device needed), from
 * Hc[k] = round(h[k] cos(theta k) 2^E) and Hs[k] = round(h[k] sin(theta k) 2^E), theta k = 2 pi ((k freg) mod 2^32) / 2^32
 * (`mix` = 0: H[k] = round(h[k] 2^E), one table, pddc_fir_i8_table's).  A table is 4 digit planes x ksteps x 64 lanes x 16
 * bytes in the matrix instruction's lane order.  hist = 32, 64 or 128 with `mix`: TWO tables of paired rows -- rows 0..7 the
 * band of eight outputs for one tap set, rows 8..15 the band of the same outputs for the other: [Hc ; Hs] meets the I
 * planes, [-Hs ; Hc] the Q planes, one set of int32 accumulators takes both, rows 0..7 are uI, rows 8..15 uQ;
 * ksteps = (56 + hist + 63) / 64.  hist = 256 with `mix`: the 16-row tables of Hc and of Hs, ksteps = (120 + hist + 63) / 64
 * (as without `mix`).  ct[0], ct[1]: the byte planes' offset constants of uI = gc * xI - gs * xQ and
 * uQ = gs * xI + gc * xQ.  Returns the number of tables written (1 or 2).                                           */
int pddc_fir_i8x_tables(const float *taps, int ntaps, int hist, int mix, uint32_t freg, int8_t *tables, size_t tables_bytes,
                        float *scale, float *ct /* [2] */);
/* ... the same for the tuned decimate-by-10 first stage (the 1.6 MS/s plan's; hist = 64, columns of 8 outputs 80 samples
 * apart, 3 k-steps): tt = c - 10 (r & 7); the taps delayed by `delay` = 0 .. 7 samples, g[k] = h[k - delay] e^{+j theta k},
 * which is how a batch whose first output does not fall on a multiple of 8 samples is put on the loaders' 8-sample groups
 * (ntaps + delay <= 64).  Two tables of 4 * 3 * 1024 bytes. */
int pddc_fir_i8x_d10_tables(const float *taps, int ntaps, int delay, uint32_t freg, int8_t *tables, size_t tables_bytes,
                            float *scale, float *ct /* [2] */);
/* ... and the fused second stage's taps: out[i] = Re g2[64 - i], out[68 + i] = Im g2[64 - i], i = 0 .. 64,
 * g2[k] = h2[k] e^{+j 8 theta k} (a first-stage output is 8 input samples); 136 floats */
int pddc_fir_i8x_taps2(const float *taps2, int ntaps2, int mix, uint32_t freg, float *out, size_t out_len);
/* Kernel selection is API state, not environment: name = "no_i8", "i8x", "i8x_pair", "i8x_plain", "i8x_blocks",
 * "i8x_chunk", "i8x_layout", "i8x_pair_max_log2", "no_fuse2", "fuse3" (the PDDC_* environment variables of the same names
 * are read once, when the pipeline is created).
 * PDDC_ESTATE while overlap mode holds a tail back (fence first), PDDC_EINVAL for an unknown name.                  */
int pddc_pipeline_set_option(pddc_pipeline *p, const char *name, int value);
/* Process-wide development knobs of the launchers (tile schedule of k_fir8, block shapes, gang plumbing): "fir8_dyn_pct",
 * "fir8_chunk", "fir8_walk" (1: chunks handed round the blocks, 0: static runs + dynamic tail, -1: the launcher's default), "gen_shape_nt", "gen_shape_p", "no_firp", "firp_packed_p", "unpack_blocks", "debug", "push_three_streams",
 * "gang_copy_out", "gang_gen_inline", "gang_solo".  Their PDDC_<NAME> environment variables are read once, at first use;
 * no library call on the data path looks at the environment.                                                         */
int pddc_set_tunable(const char *name, int value);
int pddc_get_tunable(const char *name, int *value);
int pddc_pipeline_get_option(const pddc_pipeline *p, const char *name, int *value);
/* The tap operand k_fir_i8x's plain form reads, as the library builds it (host arithmetic, no device needed; for tests and for
 * hosts that want to look at the quantisation): taps -> H[k] = round(h[k] * 2^E), E = 30 - ceil(log2 max|h|), as four
 * balanced base-256 digits d_j[k] in [-128, 127]; table[j][ks][lane][jj] = d_j[hist - (c - 8 r)] for r = lane & 15,
 * c = 64 ks + 16 (lane >> 4) + jj and 1 <= c - 8 r <= hist, else 0 (the banded Toeplitz matrix in the lane order of
 * v_mfma_i32_16x16x64_i8); 4 * ksteps * 1024 bytes with ksteps = (120 + hist + 63) / 64, hist = 32, 64, 128 or 256.  *scale turns the
 * integer result into the reference's float, *cterm is the constant that undoes the byte planes' -128 offset.       */
int pddc_fir_i8_table(const float *taps, int ntaps, int hist, int8_t *table, size_t table_bytes, float *scale, float *cterm);
/* With PDDC_F_TAPS_FP16 (no NCO) the device holds no such table but the taps as IEEE binary16, and the kernel's matrix waves quantise
 * them into their operand registers: out[128 + tt] = binary16(h[hist - tt]) for tt = 1 .. hist, zeros elsewhere,
 * PDDC_FIR_I8_TAPS16_LEN entries (lane (r, kq) reads the 16 values 128 + 64 ks + 16 kq - 8 r + jj of k-step ks as two
 * aligned 16-byte loads); H = llround(value * *two_e), digits as above.  Host arithmetic, no device needed. */
#define PDDC_FIR_I8_TAPS16_LEN 512
int pddc_fir_i8_taps16(const float *taps, int ntaps, int hist, uint16_t *out, size_t out_len, double *two_e);
/* 1 if stage 0 reads the packed samples itself (the fused decimate-by-8, or the generic decimator
 * with its unpack-while-staging load phase for any other first decimation): no float32
 * intermediate of the input is ever written; 6 + 8/D bytes per input sample                  */
int pddc_pipeline_stage0_reads_packed(const pddc_pipeline *p);
/* 1 if a process() of nsamples would run stages 0 AND 1 as one kernel (both
 * decimate-by-8, stage 1 <= 64 taps, nsamples a multiple of the kernel's tile):
 * the stage-0 output then never reaches HBM                                    */
int pddc_pipeline_uses_fused_pair(const pddc_pipeline *p, size_t nsamples);
/* 1 if a process() of nsamples would run stages 0, 1 AND 2 as one kernel (the fused pair followed by a plain
 * decimator whose group of tiles fits the LDS, e.g. the 8*8*5 and 8*8*10 plans of the 250 and 125 kS/s rates):
 * the whole cascade is then one streaming pass, 6 + 8/D bytes per input sample                               */
int pddc_pipeline_uses_fused_cascade(const pddc_pipeline *p, size_t nsamples);
/* Waits for `stream`, then reports whether a kernel of this pipeline has flagged a failure since the last reset (the
 * fused cascade's blocks hand each other FIR history inside the launch; their wait is bounded and a block that
 * gives up says so here: PDDC_EHIP).  PDDC_OK otherwise.  Tests and bench.py call it after their runs.          */
int pddc_pipeline_check(pddc_pipeline *p, void *stream);

/* Device-resident batch: d_packed (16-byte aligned, nsamples % 8 == 0) ->
 * d_out_f32 (16-byte aligned, capacity in complex samples; with
 * PDDC_F_OUT_PACKED24 it receives 6 bytes per sample instead of 8).  Asynchronous on
 * `stream`; *n_out (host) is written before return (it depends only on sizes).
 * Stream state (FIR history, phase, NCO counter) advances by nsamples.  The state
 * lives in device memory ordered by `stream`: use ONE stream per pipeline, and do
 * not mix process() and push_host() (which runs on the pipeline's own stream)
 * without a synchronisation in between.                                        */
int pddc_pipeline_process(pddc_pipeline *p, const void *d_packed, size_t nsamples,
                          void *d_out_f32, size_t out_capacity, size_t *n_out, void *stream);
/* Overlap mode for three-stage cascades whose first two stages run as the fused pair (pddc_pipeline_uses_fused_pair):
 * the third stage -- 1/64 of the samples, a tenth of the arithmetic, but one more kernel in the row with its launch
 * gaps, 11 % of the /320 step -- is NOT launched with its batch.  The NEXT process() makes it part of its own launch:
 * extra thread blocks behind the pair's persistent ones, on waves the pair leaves idle (the pair writes the other half
 * of a double-buffered workspace meanwhile).  One stream, no events.  The price is one batch of latency in what
 * `stream` holds: behind process() k, the outputs up to batch k-1 are complete; pddc_pipeline_fence(p, stream) launches
 * the tail that is still held back -- call it before the last output is read, and before save_state / set_taps /
 * set_overlap / set_workspace / set_option (they answer PDDC_ESTATE while a tail is held back).  The output buffer given to process() k must stay valid until the launch of k+1 (or the fence).
 * Batches that do not take the fused pair (not whole tiles) fence by themselves and run in line; push_host* fence every
 * batch (they are PCIe-bound).  Call set_overlap BEFORE pddc_pipeline_workspace_size / _set_workspace: the workspace
 * then holds stage 2's input twice.  No reference counterpart (the FPGA's stages all run at once).              */
int pddc_pipeline_set_overlap(pddc_pipeline *p, int enable);
int pddc_pipeline_fence(pddc_pipeline *p, void *stream);
/* Host batch: H2D copy, process, D2H copy, synchronous.                        */
int pddc_pipeline_push_host(pddc_pipeline *p, const void *h_packed, size_t nsamples,
                            void *h_out_f32, size_t out_capacity, size_t *n_out);
/* The same, returning at once: the batch travels H2D -> kernels -> D2H on three streams
 * through one of two staging slots, so with two batches in flight the copy in, the kernels
 * and the copy out of neighbouring batches overlap (the pinned, double-buffered host ring
 * the reference's 6 KB callback buffers need).  *n_out is known at
 * return (it depends on sizes only); h_packed and h_out must stay valid, and h_out unread,
 * until pddc_pipeline_wait_ticket(*ticket) -- at most two tickets (0, 1) are outstanding,
 * pushing a third batch reuses the older slot and waits for it on the device side.  For
 * the copies to be truly asynchronous both host buffers should come from pddc_host_alloc
 * (pinned memory); pageable memory works but the runtime stages it.                      */
int pddc_pipeline_push_host_async(pddc_pipeline *p, const void *h_packed, size_t nsamples,
                                  void *h_out_f32, size_t out_capacity, size_t *n_out, int *ticket);
/* The same with the synthetic source of BASELINE.md section 3 generated ON the device (the
 * batch is bytes [byte_offset, byte_offset + 6*nsamples) of the LCG stream of `seed`,
 * bit-identical to the host loop): no host -> device traffic, so a C host driving several
 * virtual receivers is not bound by a single-thread CPU generator.                         */
int pddc_pipeline_push_synth_async(pddc_pipeline *p, uint32_t seed, uint64_t byte_offset, size_t nsamples,
                                   void *h_out_f32, size_t out_capacity, size_t *n_out, int *ticket);
/* 1 if the batch of `ticket` has completely arrived in its h_out, 0 if not yet (non-blocking) */
int pddc_pipeline_ticket_done(pddc_pipeline *p, int ticket);
int pddc_pipeline_wait_ticket(pddc_pipeline *p, int ticket);
int pddc_pipeline_wait(pddc_pipeline *p);          /* everything pushed so far is complete */
/* ---- gang: several pipelines of ONE GPU, one launch chain --------------------------------
 * The reference serves up to eight receivers from one poll thread (perseus-sdr.c:43-47 the
 * descriptor table, 736-774 the thread); here several of them may share a GPU.  A batch of
 * 2^22 samples keeps the GPU busy for a few dozen microseconds -- about what ONE launch costs
 * in front of it -- so a launch chain per receiver leaves the GPU waiting for the host.  A gang
 * round pushes the next batch (the same nsamples) of up to PDDC_GANG_MAX pipelines through one
 * stream with ONE launch per kernel: the generator (or one H2D copy each), the first-stage
 * kernel with the receiver as the grid's second dimension -- every receiver its own tuning
 * word, phase, histories and buffers --, the decimator behind it likewise, one D2H copy
 * each and ONE event.  Results are bit-identical to pushing every pipeline by itself
 * (pddc_pipeline_push_*_async): the kernels' code and every receiver's arguments are the same.
 * Members whose plan is not "fused /8 first stage [+ /8 fused with it] [+ one plain
 * decimator]" (resampling rates, a first stage that is not /8, packed output) still go out in
 * the round, as launches of their own on the gang's stream; *n_ganged says how many shared.
 * Tickets are the pipelines' own (pddc_pipeline_wait_ticket).  A pipeline may change between
 * gang rounds and pushes of its own at any batch boundary (the change waits for what it still
 * has in flight).  One thread at a time per gang.  Everything that can refuse a round (null
 * pointers, capacities, a pipeline twice, a held-back overlap tail) is checked before anything is
 * queued: such a call leaves every stream where it was.  A failure AFTER that (a HIP error while
 * queueing) leaves the round's members in an undefined position: reset or destroy them.  A
 * member's h_out must stay the kind of memory it was when first seen (pinned or pageable): the
 * answer is remembered per staging slot.                                                     */
#define PDDC_GANG_MAX 8
typedef struct pddc_gang pddc_gang;
typedef struct {
    pddc_pipeline *pipe;
    const void    *h_packed;      /* host batch (6*nsamples bytes), or NULL: the on-device LCG source ... */
    uint32_t       seed;          /* ... of this seed ...                                                 */
    uint64_t       byte_offset;   /* ... from this byte of its stream                                     */
    void          *h_out;         /* receives the outputs (float2, or packed with PDDC_F_OUT_PACKED24)     */
    size_t         out_capacity;  /* in samples                                                            */
    size_t         n_out;         /* out: outputs of this batch                                            */
    int            ticket;        /* out                                                                   */
} pddc_gang_item;
int pddc_gang_create(pddc_gang **out, int device);
int pddc_gang_destroy(pddc_gang *g);      /* after (or before) its pipelines: waits for what is in flight */
int pddc_gang_push_async(pddc_gang *g, pddc_gang_item *items, int n, size_t nsamples, int *n_ganged);

/* pinned host memory for the two calls above */
int pddc_host_alloc(void **h_ptr, size_t nbytes);
int pddc_host_free(void *h_ptr);

/* ---- measurement hooks (bench.py) ----------------------------------------- */
/* Times `iters` back-to-back launches of the pipeline's stage-0 kernel alone
 * with HIP events on `stream`; returns average milliseconds per launch.
 * State is not advanced (history taken as is).                                 */
int pddc_pipeline_time_stage0(pddc_pipeline *p, const void *d_packed, size_t nsamples,
                              void *d_out_f32, int iters, void *stream, float *avg_ms);

/* The same measurement INSIDE the caller's own loop: while enabled, every process() brackets its stage-0
 * (or fused-pair) kernel with a pair of HIP events on the caller's stream; pddc_pipeline_stage0_time waits for
 * them and returns the average kernel duration over the process() calls since it was enabled (or last read).
 * This is how bench.py gets the dominant kernel's duration over exactly its timed region.                  */
int pddc_pipeline_time_stage0_inline(pddc_pipeline *p, int enable);
int pddc_pipeline_stage0_time(pddc_pipeline *p, float *avg_ms, int *nlaunches);
/* The tile schedule a process() of nsamples would launch the fused stage-0 kernel with:
 * out = { inputs per tile, tiles, persistent blocks, S, K } -- block b first owns the S
 * tiles [b*S, (b+1)*S), the tiles from blocks*S on are handed out in chunks of K.  For
 * tests and bench.py, which place their comparison windows on these seams.           */
int pddc_pipeline_schedule(const pddc_pipeline *p, size_t nsamples, int out[5]);
/* Checkpoint / resume.  The stream state the reference never needed (its FPGA kept it): per stage the
 * FIR history and the decimation phase, the sample counter, the NCO word with its phase offset.  save:
 * one host blob (pddc_pipeline_state_size bytes), taken after everything pushed so far has completed.
 * restore: into a pipeline of the SAME plan and flags, on the same or another GPU -- the stream then
 * continues bit-identically (a receiver can move between GPUs, or survive a restart).             */
size_t pddc_pipeline_state_size(const pddc_pipeline *p);
int pddc_pipeline_save_state(pddc_pipeline *p, void *h_buf, size_t capacity, size_t *used);
int pddc_pipeline_restore_state(pddc_pipeline *p, const void *h_buf, size_t nbytes);
/* Test hook: the next process()/push fails with PDDC_EHIP when it reaches `stage`, after the stages
 * in front of it have been launched -- to show that a failure half way leaves the stream state
 * (histories, decimation phases, NCO counter) where it was and the batch can simply be retried.   */
int pddc_pipeline_inject_failure(pddc_pipeline *p, int stage);
/* Device-to-device streaming copy of nbytes (16 B per lane, nontemporal stores), `iters`
 * times, timed with HIP events on `stream`: the measured copy ceiling bench.py prints
 * next to the 8 TB/s spec figure (SURVEY.md 8d "Which roofline").  A copy moves
 * 2*nbytes through HBM.  Pointers and size: multiples of 16.                         */
int pddc_measure_copy(void *d_dst, const void *d_src, size_t nbytes, int iters, void *stream, float *avg_ms);

/* ---- multi-GPU: RCCL over xGMI (ddc_multi.cpp) --------------------------------
 * The reference models up to 8 receivers as 8 independent descriptors, each with
 * its own transfer queue and callback (perseus-sdr.c:43-47, perseus-in.h:87): the
 * stream shards as one receiver per GPU and the data path needs no collective.
 * What these calls carry between GPUs is the configuration (root -> all) and,
 * for BASELINE config 4, every GPU's decimated output to one root GPU.  They
 * stand where the reference has nothing (its receivers never talk to each other);
 * the hand-over point to the client stays perseus-in.c:206-207.
 *
 * One communicator rank per GPU.  pddc_comm_init_rank: one process per GPU -- rank 0
 * calls pddc_comm_get_unique_id and hands the 128 bytes to the others through
 * whatever started the processes (bench.py: the torch.distributed store).
 * pddc_comm_init_all: ONE process driving ndev GPUs (a C host holding several
 * perseus_descr); its collective calls -- one per communicator, same arguments --
 * go between pddc_comm_group_start() and pddc_comm_group_end().
 * All sizes in bytes.  Errors: PDDC_ECOMM + pddc_last_error().                  */
typedef struct pddc_comm pddc_comm;
#define PDDC_COMM_ID_BYTES 128
/* Version codes (ncclGetVersion style: major*10000 + minor*100 + patch) of the RCCL library bound at run time and of the
 * header this library was compiled against.  Every communicator constructor checks that the MAJOR versions agree and
 * fails with PDDC_ECOMM otherwise (two librccl can be on a box: /opt/rocm/lib and the one inside a torch wheel).   */
int pddc_comm_rccl_version(int *running, int *compiled);
int pddc_comm_get_unique_id(void *id128);
int pddc_comm_init_rank(pddc_comm **out, int nranks, int rank, const void *id128, int device);
int pddc_comm_init_all(pddc_comm **comms /* [ndev] */, int ndev, const int *devices /* NULL: 0..ndev-1 */);
int pddc_comm_destroy(pddc_comm *c);
int pddc_comm_rank(const pddc_comm *c);
int pddc_comm_size(const pddc_comm *c);
int pddc_comm_device(const pddc_comm *c);
int pddc_comm_group_start(void);
int pddc_comm_group_end(void);
/* root's d_buf -> everybody's d_buf (ncclBroadcast), asynchronous on `stream`     */
int pddc_comm_bcast(pddc_comm *c, void *d_buf, size_t nbytes, int root, void *stream);
/* the same for a host buffer, synchronous (one process per GPU only)              */
int pddc_comm_bcast_host(pddc_comm *c, void *h_buf, size_t nbytes, int root);
/* *h_val = max over ranks; pddc_comm_barrier = the same with nothing to say        */
int pddc_comm_allreduce_max_f64(pddc_comm *c, double *h_val);
int pddc_comm_barrier(pddc_comm *c);
/* Gather: every rank's nbytes at d_send land on the root at d_recv + rank*nbytes
 * (d_recv is read on the root only).  Grouped ncclSend/ncclRecv, peer -> root: the
 * root's xGMI links carry one peer each.  Asynchronous on `stream`.                */
int pddc_comm_gather(pddc_comm *c, const void *d_send, size_t nbytes, void *d_recv, int root, void *stream);
/* The same on the communicator's own side stream, started once everything queued on
 * `after_stream` so far (the kernels that wrote d_send) is done -- the transfer of
 * batch k then runs under the kernels of batch k+1.  Meant for TWO alternating send
 * buffers: pddc_comm_gather_fence(c, stream) makes `stream` wait for every transfer
 * but the most recent one -- call it before the kernels that overwrite the buffer
 * used two gathers ago; pddc_comm_gather_wait(c) makes the host wait for all of them
 * (before d_recv is read, or a buffer is freed).                                    */
int pddc_comm_gather_async(pddc_comm *c, const void *d_send, size_t nbytes, void *d_recv, int root,
                           void *after_stream);
int pddc_comm_gather_fence(pddc_comm *c, void *stream);
int pddc_comm_gather_wait(pddc_comm *c);

/* The configuration a broadcast carries: stage plan + taps + NCO word + flags, as one
 * flat little-endian buffer.  pack: returns the bytes used (buf == NULL: the bytes
 * needed), 0 on error.  unpack: stages[i].taps point INTO buf.                      */
size_t pddc_plan_pack(const pddc_stage_desc *stages, int nstages, uint32_t freg, uint32_t flags,
                      void *buf, size_t capacity);
int pddc_plan_unpack(const void *buf, size_t nbytes, pddc_stage_desc *stages /* [PDDC_MAX_STAGES] */,
                     int *nstages, uint32_t *freg, uint32_t *flags);
/* Root supplies the plan (others pass NULL/0); every rank gets a pipeline on its
 * communicator's GPU, created from the broadcast plan, NCO word set.                */
int pddc_comm_bcast_pipeline(pddc_comm *c, int root, const pddc_stage_desc *stages, int nstages,
                             uint32_t freg, uint32_t flags, pddc_pipeline **out);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* PERSEUS_DDC_H */
