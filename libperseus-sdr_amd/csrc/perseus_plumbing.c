/*
 * perseus_plumbing.c -- command-line client for the plumbing check of
 * BASELINE.json configs[0]: the call sequence of the reference's
 * examples/perseustest.c:188-404 (init, open, firmware, product id, rate,
 * attenuator, ADC, tuning, start, wait, stop, close, exit) against this
 * repository's libperseus-sdr.so.
 *
 * In wire mode the library hands over 24-bit packed samples exactly as the
 * reference does, and -- as in the reference, where the sample conversion lives
 * in the example client, not in the library (perseustest.c:432-502) -- this
 * client's callbacks convert them on the CPU.  In DDC mode (PERSEUS_AMD_MODE=ddc)
 * the buffers already hold float32 I/Q from the GPU and are written as is.
 *
 *   -s rate   sampling rate in S/s (default 95000)     -n nb -b bs  buffer = nb*bs bytes (6*1024)
 *   -f hz     tuning frequency (7000000)               -a           no attenuator/ADC test calls
 *   -t sec    run time (10)                            -m count     stop after count buffers
 *   -o file   output ("perseusdata", "-" = stdout)     -p           float32 output (default int32)
 *   -d level  debug level (3)                          -F fifo      control FIFO (see below)
 *   -B w,k    bench mode (one receiver, DDC mode): stop after w + k + 1 GPU batches and report the time the callbacks took
 *             from the end of batch w to the end of batch w + k (stamps taken in the callback): "api_bench ..." line
 *   -N n      open n receivers (1..8, reference PERSEUS_MAX_DESCR; sets PERSEUS_AMD_DEVICES=n unless it is set):
 *             every receiver gets the same settings and its own stream (seed 12345+i); in DDC mode receiver i
 *             runs on GPU i % ngpu and all of them are in flight at once; output files get ".i" appended
 *
 * -F creates a named pipe and a control thread, as the reference example's fifo.c
 * does: each line is "<MHz as float>", "<Hz as integer>", "att <0..3>" or "quit";
 * tuning and attenuator calls are made WHILE streaming (examples/fifo.c:23-60) and
 * in DDC mode a retune takes effect at the next GPU batch boundary.
 */
#include "../../include/perseus-amd-ext.h"

#include <fcntl.h>
#include <pthread.h>
#include <stdatomic.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <time.h>
#include <unistd.h>

typedef struct {
    FILE *out;
    int ddc;
    unsigned long long buffers, samples;
    /* -B: time stamps taken in the callback when the delivered samples cross two marks (bench.py --workload api250k) */
    unsigned long long mark0, mark1;
    struct timespec t_mark0, t_mark1;
    int got0, got1;
} sink;

static inline void sink_marks(sink *s)
{
    if (s->mark1 == 0)
        return;
    if (!s->got0 && s->samples >= s->mark0) {
        clock_gettime(CLOCK_MONOTONIC, &s->t_mark0);
        s->got0 = 1;
    }
    if (!s->got1 && s->samples >= s->mark1) {
        clock_gettime(CLOCK_MONOTONIC, &s->t_mark1);
        s->got1 = 1;
    }
}

/* wire format: I0 I1 I2 Q0 Q1 Q2, 24-bit little endian.  MSB-align into int32. */
static inline int32_t msb24(const uint8_t *b)
{
    return (int32_t)(((uint32_t)b[0] << 8) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 24));
}

static int on_buffer_int32(void *buf, int buf_size, void *extra)
{
    sink *s = (sink *)extra;
    const uint8_t *b = (const uint8_t *)buf;
    const int n = buf_size / 6;
    int32_t iq[2 * 2720];
    for (int k = 0; k < n; k++) {
        iq[2 * k] = msb24(b + 6 * k);
        iq[2 * k + 1] = msb24(b + 6 * k + 3);
    }
    if (s->out)
        fwrite(iq, sizeof(int32_t), 2 * (size_t)n, s->out);
    s->buffers++;
    s->samples += (unsigned)n;
    return 0;
}

static int on_buffer_float(void *buf, int buf_size, void *extra)
{
    sink *s = (sink *)extra;
    if (s->ddc) {                       /* already float32 I/Q from the GPU */
        if (s->out)
            fwrite(buf, 1, (size_t)buf_size, s->out);
        s->buffers++;
        s->samples += (unsigned)buf_size / 8;
        sink_marks(s);
        return 0;
    }
    const uint8_t *b = (const uint8_t *)buf;
    const int n = buf_size / 6;
    const float full_scale = (float)(INT_MAX - 256);
    float iq[2 * 2720];
    for (int k = 0; k < n; k++) {
        iq[2 * k] = (float)msb24(b + 6 * k) / full_scale;
        iq[2 * k + 1] = (float)msb24(b + 6 * k + 3) / full_scale;
    }
    if (s->out)
        fwrite(iq, sizeof(float), 2 * (size_t)n, s->out);
    s->buffers++;
    s->samples += (unsigned)n;
    return 0;
}

static perseus_descr *g_descr;
static atomic_int g_quit;

static void *fifo_thread(void *arg)
{
    const char *path = (const char *)arg;
    char line[128];
    while (!g_quit) {
        FILE *f = fopen(path, "r");              /* blocks until a writer shows up */
        if (!f)
            break;
        while (!g_quit && fgets(line, sizeof(line), f)) {
            int n;
            if (strncmp(line, "quit", 4) == 0) {
                g_quit = 1;
            } else if (sscanf(line, "att %d", &n) == 1) {
                perseus_set_attenuator_n(g_descr, n);
            } else if (strchr(line, '.')) {
                perseus_set_ddc_center_freq(g_descr, atof(line) * 1e6, 1);
            } else if (atol(line) > 0) {
                perseus_set_ddc_center_freq(g_descr, (double)atol(line), 1);
            }
        }
        fclose(f);
    }
    return NULL;
}

#define MAX_RX 8

int main(int argc, char **argv)
{
    int rate = 95000, nb = 6, bs = 1024, dbg = 3, seconds = 10, as_float = 0, test_fe = 1, nrx = 1, multi = 0;
    long max_buffers = 0;
    int bench_w = -1, bench_k = 0;
    double freq = 7000000.0;
    const char *outname = "perseusdata";
    const char *fifo = NULL;
    pthread_t fifo_tid;
    int c;
    while ((c = getopt(argc, argv, "s:n:b:d:t:o:f:m:F:N:B:pah")) != -1) {
        switch (c) {
        case 's': rate = atoi(optarg); break;
        case 'n': nb = atoi(optarg); break;
        case 'b': bs = atoi(optarg); break;
        case 'd': dbg = atoi(optarg); break;
        case 't': seconds = atoi(optarg); break;
        case 'o': outname = optarg; break;
        case 'f': freq = atof(optarg); break;
        case 'm': max_buffers = atol(optarg); break;
        case 'F': fifo = optarg; break;
        case 'N': nrx = atoi(optarg); multi = 1; break;   /* -N 1: the same loop with one receiver */
        case 'B':
            if (sscanf(optarg, "%d,%d", &bench_w, &bench_k) != 2 || bench_w < 0 || bench_k < 1) {
                fprintf(stderr, "-B wants warm,steps\n");
                return 2;
            }
            break;
        case 'p': as_float = 1; break;
        case 'a': test_fe = 0; break;
        default:
            fprintf(stderr, "usage: %s [-s rate] [-n nb] [-b bs] [-f hz] [-t sec] [-m buffers] [-o file] [-p] [-a] [-d dbg]\n",
                    argv[0]);
            return 2;
        }
    }
    if (nrx < 1 || nrx > MAX_RX) {
        fprintf(stderr, "-N wants 1..%d receivers\n", MAX_RX);
        return 2;
    }
    if (nrx > 1) {
        char v[8];
        snprintf(v, sizeof(v), "%d", nrx);
        setenv("PERSEUS_AMD_DEVICES", v, 0);
    }
    perseus_set_debug(dbg);
    int rates[16];
    if (perseus_get_sampling_rates(NULL, rates, 16) == 0) {
        fprintf(stderr, "Sampling rates:");
        for (int i = 0; i < 16 && rates[i]; i++)
            fprintf(stderr, " %d", rates[i]);
        fprintf(stderr, "\n");
    }
    const int ndev = perseus_init();
    fprintf(stderr, "%d Perseus receivers found\n", ndev);
    if (ndev <= 0) {
        perseus_exit();
        return 1;
    }
    if (multi) {
        /* several receivers at once: same call sequence per receiver, one callback sink each */
        if (ndev < nrx) {
            fprintf(stderr, "only %d receivers present, %d wanted\n", ndev, nrx);
            perseus_exit();
            return 1;
        }
        perseus_descr *rx[MAX_RX];
        sink sk[MAX_RX];
        struct timeval t0, t1;
        for (int i = 0; i < nrx; i++) {
            rx[i] = perseus_open(i);
            if (!rx[i] || perseus_firmware_download(rx[i], NULL) < 0 || perseus_set_sampling_rate(rx[i], rate) < 0) {
                fprintf(stderr, "receiver %d: %s\n", i, perseus_errorstr());
                perseus_exit();
                return 1;
            }
            perseus_set_ddc_center_freq(rx[i], freq, 1);
            perseus_amd_config cfg;
            perseus_amd_get_config(rx[i], &cfg);
            if (max_buffers > 0) {
                cfg.max_buffers = (uint64_t)max_buffers;
                perseus_amd_set_config(rx[i], &cfg);
            }
            sk[i] = (sink){ NULL, cfg.mode == PERSEUS_AMD_MODE_DDC, 0, 0, 0, 0, { 0, 0 }, { 0, 0 }, 0, 0 };
            if (strcmp(outname, "none") != 0 && strcmp(outname, "-") != 0) {
                char name[1100];
                snprintf(name, sizeof(name), "%s.%d", outname, i);
                sk[i].out = fopen(name, "wb");
            }
        }
        gettimeofday(&t0, NULL);
        for (int i = 0; i < nrx; i++)
            if (perseus_start_async_input(rx[i], (uint32_t)(nb * bs), (as_float || sk[i].ddc) ? on_buffer_float : on_buffer_int32,
                                          &sk[i]) < 0) {
                fprintf(stderr, "receiver %d: start async input error: %s\n", i, perseus_errorstr());
                perseus_exit();
                return 1;
            }
        fprintf(stderr, "Collecting input samples from %d receivers... \n", nrx);
        for (int t = 0; t < seconds * 100; t++) {
            int running = 0;
            for (int i = 0; i < nrx; i++)
                running += perseus_amd_source_running(rx[i]);
            if (!running)
                break;
            usleep(10000);
        }
        gettimeofday(&t1, NULL);
        perseus_amd_stats st;
        unsigned long long total = 0, adc = 0;
        for (int i = 0; i < nrx; i++) {
            perseus_amd_get_stats(rx[i], &st);
            adc += st.adc_samples;
            perseus_stop_async_input(rx[i]);
            if (sk[i].out)
                fclose(sk[i].out);
            fprintf(stderr, "receiver %d (GPU %d): %llu buffers, %llu samples, %llu GPU batches (%llu in shared launches)\n", i,
                    st.gpu_device, sk[i].buffers, sk[i].samples, (unsigned long long)st.batches,
                    (unsigned long long)st.ganged_batches);
            total += sk[i].samples;
        }
        const double el = (t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_usec - t0.tv_usec);
        fprintf(stderr, "%d receivers: %llu samples in %.3f s = %.1f kS/s aggregate", nrx, total, el, total / el / 1e3);
        if (adc)
            fprintf(stderr, " (%.1f MS/s of ADC-rate input through the GPUs; most receivers with a batch in flight "
                            "at once: %d)", adc / el / 1e6, st.peak_receivers_in_flight);
        fprintf(stderr, "\n");
        perseus_exit();
        fprintf(stderr, "Bye\n");
        return 0;
    }
    perseus_descr *d = perseus_open(0);
    if (!d) {
        fprintf(stderr, "open error: %s\n", perseus_errorstr());
        perseus_exit();
        return 1;
    }
    if (perseus_firmware_download(d, NULL) < 0) {
        fprintf(stderr, "firmware download error: %s\n", perseus_errorstr());
        return 1;
    }
    eeprom_prodid id;
    if (perseus_get_product_id(d, &id) == 0)
        fprintf(stderr, "Receiver S/N: %05d-%02hX%02hX-%02hX%02hX-%02hX%02hX - HW Release:%hd.%hd\n", id.sn,
                (unsigned short)id.signature[5], (unsigned short)id.signature[4], (unsigned short)id.signature[3],
                (unsigned short)id.signature[2], (unsigned short)id.signature[1], (unsigned short)id.signature[0],
                (short)id.hwrel, (short)id.hwver);
    if (perseus_set_sampling_rate(d, rate) < 0) {
        fprintf(stderr, "fpga configuration error: %s\n", perseus_errorstr());
        return 1;
    }
    if (test_fe) {
        /* the reference cycles the relays with sleeps and feeds one bad value on purpose */
        perseus_set_attenuator_in_db(d, 33);            /* bad value: PERSEUS_ATTERROR */
        perseus_set_attenuator_n(d, 3);
        perseus_set_attenuator(d, PERSEUS_ATT_0DB);
        perseus_set_adc(d, 1, 0);
        perseus_set_ddc_center_freq(d, freq, 0);
        perseus_set_attenuator_in_db(d, 30);
    }
    perseus_set_ddc_center_freq(d, freq, 1);

    perseus_amd_config cfg;
    perseus_amd_get_config(d, &cfg);
    if (max_buffers > 0) {
        cfg.max_buffers = (uint64_t)max_buffers;
        perseus_amd_set_config(d, &cfg);
    }
    sink s = { NULL, cfg.mode == PERSEUS_AMD_MODE_DDC, 0, 0, 0, 0, { 0, 0 }, { 0, 0 }, 0, 0 };
    unsigned long long bench_batch = 0;
    if (bench_w >= 0) {
        if (!s.ddc) {
            fprintf(stderr, "-B needs PERSEUS_AMD_MODE=ddc\n");
            return 2;
        }
        /* the library's own batch size (PERSEUS_AMD_BATCH, or what it picks for this kind of source) */
        bench_batch = perseus_amd_effective_batch(d);
        const unsigned long long per_batch = bench_batch * (unsigned long long)rate / 80000000ull;   /* outputs per batch */
        s.mark0 = per_batch * (unsigned long long)bench_w;
        s.mark1 = per_batch * (unsigned long long)(bench_w + bench_k);
        cfg.max_buffers = (per_batch * (unsigned long long)(bench_w + bench_k + 1) * 8 + (unsigned long long)(nb * bs) - 1) /
                          (unsigned long long)(nb * bs);
        perseus_amd_set_config(d, &cfg);
        if (s.mark0 == 0)
            clock_gettime(CLOCK_MONOTONIC, &s.t_mark0), s.got0 = 1;
    }
    if (strcmp(outname, "-") == 0)
        s.out = stdout;
    else if (strcmp(outname, "none") != 0)
        s.out = fopen(outname, "wb");

    const int rc = perseus_start_async_input(d, (uint32_t)(nb * bs),
                                             (as_float || s.ddc) ? on_buffer_float : on_buffer_int32, &s);
    if (rc < 0) {
        fprintf(stderr, "start async input error: %s\n", perseus_errorstr());
        perseus_close(d);
        perseus_exit();
        return 1;
    }
    g_descr = d;
    if (fifo) {
        mkfifo(fifo, 0600);
        pthread_create(&fifo_tid, NULL, fifo_thread, (void *)fifo);
    }
    fprintf(stderr, "Collecting input samples... \n");
    for (int t = 0; t < seconds * 100; t++) {
        if (!perseus_amd_source_running(d) || g_quit)
            break;
        usleep(10000);
    }
    fprintf(stderr, "done\n");
    perseus_stop_async_input(d);
    if (fifo) {
        g_quit = 1;
        int k = open(fifo, O_WRONLY | O_NONBLOCK);  /* unblock a reader waiting in fopen() */
        if (k >= 0)
            close(k);
        pthread_join(fifo_tid, NULL);
        unlink(fifo);
    }
    fprintf(stderr, "final NCO word: %u\n", perseus_amd_get_freg(d));
    if (s.out && s.out != stdout)
        fclose(s.out);
    fprintf(stderr, "%llu buffers, %llu samples\n", s.buffers, s.samples);
    if (bench_w >= 0) {
        perseus_amd_stats st;
        perseus_amd_get_stats(d, &st);
        if (s.got0 && s.got1) {
            const double ms = 1e3 * (double)(s.t_mark1.tv_sec - s.t_mark0.tv_sec) + 1e-6 * (double)(s.t_mark1.tv_nsec - s.t_mark0.tv_nsec);
            printf("api_bench batch_samples=%llu warm=%d steps=%d ms_total=%.6f ms_per_step=%.6f adc_MSps=%.1f gpu_batches=%llu "
                   "ganged=%llu gpu_source=%d\n", bench_batch, bench_w, bench_k, ms, ms / bench_k,
                   (double)bench_batch * bench_k / ms / 1e3, (unsigned long long)st.batches, (unsigned long long)st.ganged_batches,
                   st.gpu_source);
        } else {
            printf("api_bench failed: the stream ended before the marks (%llu samples of %llu)\n", s.samples, s.mark1);
        }
    }
    perseus_close(d);
    perseus_exit();
    fprintf(stderr, "Bye\n");
    return 0;
}
