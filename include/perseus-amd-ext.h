/*
 * perseus-amd-ext.h -- what this build adds to the reference API.
 *
 * The reference talks to a USB receiver whose FPGA does the DSP.  Here the
 * USB side is a *source* (synthetic or file) and the FPGA's work runs on the
 * GPU, so two things need configuring that the reference API has no call for:
 * where samples come from, and what the callback buffers carry.
 *
 * Everything can also be set through environment variables read by
 * perseus_init(), so an unmodified reference client can be pointed at a source:
 *   PERSEUS_AMD_DEVICES   number of virtual receivers, 1..8 (default 1; the
 *                         reference's PERSEUS_MAX_DESCR is 8, perseus-sdr.c:43)
 *   PERSEUS_AMD_MODE      "wire" (default) | "ddc" | "ddc-wire"
 *   PERSEUS_AMD_SOURCE    "lcg[:seed]" (default lcg:12345) | "zero" | "file:<path>"
 *   PERSEUS_AMD_PACE      1 = pace the source at the nominal rate (default), 0 = free-running
 *   PERSEUS_AMD_BATCH     ddc mode: ADC-rate samples per GPU batch (default 2^22)
 *   PERSEUS_AMD_DROP      fault injection: every k-th transfer completes short
 *                         and is dropped like perseus-in.c:209-216 (default 0 = never)
 *   PERSEUS_AMD_FAULTS    fault script, e.g. "timeout@9,oos@12,error@20,short%7,eof@500"
 *                         (see perseus_amd_config.fault_script)
 *   PERSEUS_AMD_EP_PACKET endpoint max packet size the receiver reports: 512 (default) | 510
 *   PERSEUS_AMD_CPU_SOURCE  DDC modes: 1 = generate the LCG stream on the host (default: on the GPU)
 *   PERSEUS_AMD_MAX_BUFFERS  the source ends after this many transfers (default 0 = unbounded)
 */
#ifndef PERSEUS_AMD_EXT_H
#define PERSEUS_AMD_EXT_H

#include "perseus-sdr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* What the callback `buf` carries.
 *  WIRE: the source plays the receiver: 24-bit packed I/Q, 6 bytes/sample, at
 *        the rate chosen with perseus_set_sampling_rate(); the library does no
 *        arithmetic, the client unpacks (examples/perseustest.c:466-502).  No GPU
 *        needed.  This is what every reference client expects.
 *  DDC:  the source delivers the 80 MS/s ADC-rate 24-bit I/Q stream; the GPU
 *        pipeline mixes it with the NCO set by perseus_set_ddc_center_freq()
 *        and decimates to the selected rate; callbacks carry interleaved
 *        float32 I/Q (8 bytes/sample) in buffers of `buffersize` bytes.  Needs a
 *        GPU: perseus_start_async_input() fails (no CPU fallback) without one. */
#define PERSEUS_AMD_MODE_WIRE 0
#define PERSEUS_AMD_MODE_DDC  1
/*  DDC_WIRE: as DDC, but the GPU re-quantises its output to the 24-bit wire
 *        format (pddc_pack24_f32), so callbacks look exactly like the hardware's:
 *        6 bytes/sample at the selected rate ("FPGA emulation").
 *        An unmodified reference client runs on the GPU path this way.            */
#define PERSEUS_AMD_MODE_DDC_WIRE 2

#define PERSEUS_AMD_SRC_LCG   0
#define PERSEUS_AMD_SRC_ZERO  1
#define PERSEUS_AMD_SRC_FILE  2

typedef struct {
    int         mode;         /* PERSEUS_AMD_MODE_*                              */
    int         source;       /* PERSEUS_AMD_SRC_*                               */
    uint32_t    lcg_seed;     /* LCG: s=s*1664525+1013904223, byte=s>>24          */
    const char *file_path;    /* FILE: raw 24-bit packed capture, no header       */
    int         pace;         /* 1: real-time pacing at the nominal sample rate   */
    int         gpu_device;   /* DDC: HIP device index (-1: descriptor index % n) */
    uint32_t    batch_samples;/* DDC: ADC-rate samples per GPU batch: 0 = the library picks per stream (the default;
                                 perseus_amd_effective_batch says what), else the client's choice: a multiple
                                 of 8, at most PERSEUS_AMD_BATCH_MAX */
    int         drop_every;   /* fault injection, 0 = off                         */
    uint64_t    max_buffers;  /* stop the source after this many transfers (0 = unbounded;
                                 a FILE source also stops at end of file)          */
    int         ep_packet_size;/* what the data endpoint reports as its max packet size: 512 (default)
                                 or 510 -- buffer sizes must then be multiples of 6144 or of 510
                                 bytes; anything else makes start fail like the reference
                                 (perseus-sdr.c:664-680)                            */
    int         cpu_source;   /* DDC modes, LCG source: 1 = generate on the host and copy in; 0 (default)
                                 = generate on the GPU (same bytes, no host->device traffic) */
    const char *fault_script; /* what the virtual USB side does wrong, reference semantics
                                 (perseus-in.c:199-257): "kind@n" at transfer n, "kind%k" every k-th,
                                 comma separated; kinds short, timeout, oos, error, stall, nodev,
                                 overflow, eof.  NULL = no faults                    */
} perseus_amd_config;

typedef struct {
    uint64_t delivered;       /* callbacks made                                    */
    uint64_t dropped;         /* short or out-of-sequence transfers                */
    uint64_t timeouts;        /* tolerated, resubmitted                            */
    uint64_t dead_transfers;  /* killed by a fatal status; 8 = queue completed     */
    uint64_t transfers;       /* completions seen by the dispatcher                */
    uint64_t bytes_received;
    uint64_t adc_samples;     /* DDC modes: ADC-rate samples handed to the GPU     */
    uint64_t batches;
    int      gpu_device;      /* -1: no GPU pipeline                               */
    int      gpu_source;      /* 1: the synthetic stream is generated on the GPU   */
    int      peak_receivers_in_flight;   /* library-wide since perseus_init(): the most receivers that had
                                 a GPU batch in flight at the same moment (the delivery thread submits
                                 for all of them before it waits for any)           */
    uint64_t ganged_batches;  /* batches of this receiver that shared their kernel launches with other
                                 receivers on the same GPU (pddc_gang_push_async)   */
    uint64_t buffers_in_place; /* DDC modes: transfers whose callback read the output where the GPU had put it
                                 (no host copy), */
    uint64_t buffers_gathered; /* ... and transfers copied into their ring slot first: one that straddles two batches,
                                 or every one of a stream whose batches are small against the transfers */
} perseus_amd_stats;

/* valid between perseus_open() and perseus_start_async_input() */
int perseus_amd_get_config(perseus_descr *descr, perseus_amd_config *cfg);
int perseus_amd_set_config(perseus_descr *descr, const perseus_amd_config *cfg);

/* The library's picks, and the largest batch a client may ask for (a batch is a pinned host buffer of 6 bytes a sample
 * and a device buffer of the same size, twice: 2^28 samples are 1.6 GB each) */
#define PERSEUS_AMD_BATCH_DEFAULT        (1u << 22)     /* 52 ms of signal: a paced, real-time source must not wait longer */
#define PERSEUS_AMD_BATCH_UNPACED_DEVICE (1u << 24)     /* a free-running on-device source: only the launch chain to amortise */
#define PERSEUS_AMD_BATCH_MAX            (1u << 28)
/* The GPU batch size the next stream will use: cfg.batch_samples if the client chose one (set_config, set_batch,
 * PERSEUS_AMD_BATCH); otherwise (cfg.batch_samples == 0) the library's pick for the kind of source */
uint32_t perseus_amd_effective_batch(perseus_descr *descr);
/* The same choice as cfg.batch_samples without a get_config / set_config round: batch_samples > 0 (a multiple of 8, at most
 * PERSEUS_AMD_BATCH_MAX) is the client's batch size from the next stream on; 0 hands the choice back to the library.  The
 * effective size of a stream never writes cfg.batch_samples, so a descriptor that streamed unpaced at 2^24 starts its
 * next, paced stream at 2^22 again.  Not while streaming (PERSEUS_ASYNCSTARTED). */
int perseus_amd_set_batch(perseus_descr *descr, uint32_t batch_samples);

/* state introspection (for tests and tools) */
uint32_t perseus_amd_get_freg(perseus_descr *descr);          /* NCO word, perseus-sdr.c:584 */
int      perseus_amd_get_sampling_rate(perseus_descr *descr); /* selected rate in S/s, 0 if none */
int      perseus_amd_get_frontendctl(perseus_descr *descr);   /* atten_id<<4 | presel_id   */
int      perseus_amd_get_sioctl(perseus_descr *descr);        /* FIFOEN|DITHER|GAINHIGH bits */
uint64_t perseus_amd_buffers_delivered(perseus_descr *descr);
uint64_t perseus_amd_buffers_dropped(perseus_descr *descr);
/* 1 while the source still has data (a bounded source ends by itself) */
int      perseus_amd_source_running(perseus_descr *descr);
int      perseus_amd_get_stats(perseus_descr *descr, perseus_amd_stats *st);
/* DDC modes: the tuning-word segments of the current (or last) stream: segment i starts at ADC
 * sample first_sample[i] with NCO word word[i]; a retune made while streaming takes effect at a
 * GPU batch boundary, sample-accurately and phase-continuously.  Returns the number of segments
 * (fills at most `capacity`).                                                              */
int      perseus_amd_get_retune_log(perseus_descr *descr, uint64_t *first_sample, uint32_t *word, int capacity);

/* DDC mode: the decimation plan chosen for the selected rate.  Returns the
 * number of stages, fills decim[]/ntaps[]
 * (up to 4) and, if taps[i] is non-NULL, copies stage i's taps (ntaps[i] floats). */
int perseus_amd_get_plan(perseus_descr *descr, int decim[4], int ntaps[4], float *taps[4]);
/* interpolation factor L of each stage (1 = plain decimator; >1 = rational L/decim
 * resampler, used by the 48k/95k/96k/192k plans) */
int perseus_amd_get_plan_interp(perseus_descr *descr, int interp[4]);
/* The same for a rate, without a descriptor and without touching the library's global state (no init / exit): the plan
 * perseus_set_sampling_rate(sps) would select (nearest table rate, perseus-sdr.c:776-811; *rate gets it).  Returns the
 * number of stages, or PERSEUS_FPGANOTCFGD for a rate the table does not reach.  Does not set perseus_error. */
int perseus_amd_plan_for_rate(int sps, int *rate, int decim[4], int ntaps[4], int interp[4], float *taps[4]);

#ifdef __cplusplus
}
#endif
#endif
