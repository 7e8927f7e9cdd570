/*
 * perseus_multi_bench.c -- BASELINE config 4 from a plain C host: one process, N GPUs, one
 * independent 80 MS/s-style stream per GPU (the reference models up to 8 receivers as 8
 * descriptors, perseus-sdr.c:43-47), every GPU running the decimation pipeline on its own
 * device-resident synthetic batch, and -- with -G -- every GPU's decimated output gathered on
 * GPU 0 over xGMI by the library's own RCCL calls (pddc_comm_init_all + grouped
 * pddc_comm_gather_async), batch k's transfer under batch k+1's kernels.
 *
 * Only include/perseus_ddc.h is used: no Python, no torch, no MPI.
 *
 *   -g n      GPUs to use (default: all visible)          -n log2   samples per GPU per step (26)
 *   -s steps  timed steps (50)                            -w warm   warm-up steps (5)
 *   -T file   first-stage taps, raw float32 (default ../tests/golden/taps_d8_127.f32 next to the binary)
 *   -c        cascade /320 with NCO (taps c320_* from the same directory) instead of the single /8
 *   -G        gather the outputs on GPU 0
 *   -A GiB    size of the per-GPU arena that input, workspace and outputs are cut from (72 = a quarter of the HBM; 0: separate allocations)
 * Prints one line per run: aggregate input MS/s (kernel path only), and with -G the with-gather rate
 * and the bytes per second that reached the root per peer link.
 */
#define _GNU_SOURCE
#include "../../include/perseus_ddc.h"

#include <libgen.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <unistd.h>

#define MAXG 8

static double now_s(void)
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return tv.tv_sec + 1e-6 * tv.tv_usec;
}

static float *load_f32(const char *dir, const char *name, int *n)
{
    char path[2048];
    snprintf(path, sizeof(path), "%s/%s", dir, name);
    FILE *f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "cannot open %s\n", path);
        return NULL;
    }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    float *t = (float *)malloc((size_t)sz);
    if (!t || fread(t, 1, (size_t)sz, f) != (size_t)sz) {
        fclose(f);
        free(t);
        return NULL;
    }
    fclose(f);
    *n = (int)(sz / 4);
    return t;
}

#define CHECK(call)                                                                    \
    do {                                                                               \
        int rc__ = (call);                                                             \
        if (rc__ < 0) {                                                                \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc__, pddc_last_error());   \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

int main(int argc, char **argv)
{
    int ng = 0, log2n = 26, steps = 50, warm = 5, cascade = 0, gather = 0, arena_gib = 72, c;
    char tapdir[1024];
    {
        char self[1024];
        ssize_t k = readlink("/proc/self/exe", self, sizeof(self) - 1);
        self[k > 0 ? k : 0] = 0;
        snprintf(tapdir, sizeof(tapdir), "%s/../tests/golden", dirname(self));
    }
    while ((c = getopt(argc, argv, "g:n:s:w:T:A:cGh")) != -1) {
        switch (c) {
        case 'g': ng = atoi(optarg); break;
        case 'n': log2n = atoi(optarg); break;
        case 's': steps = atoi(optarg); break;
        case 'w': warm = atoi(optarg); break;
        case 'T': snprintf(tapdir, sizeof(tapdir), "%s", optarg); break;
        case 'A': arena_gib = atoi(optarg); break;
        case 'c': cascade = 1; break;
        case 'G': gather = 1; break;
        default:
            fprintf(stderr, "usage: %s [-g gpus] [-n log2n] [-s steps] [-w warmup] [-T tapdir] [-A arena GiB] [-c] [-G]\n", argv[0]);
            return 2;
        }
    }
    const int have = pddc_device_count();
    if (have <= 0) {
        fprintf(stderr, "no GPU visible: this program has no CPU path\n");
        return 1;
    }
    if (ng <= 0 || ng > have)
        ng = have;
    if (ng > MAXG)
        ng = MAXG;
    const size_t ns = (size_t)1 << log2n;

    /* the plan: one /8 stage, or the /320 cascade with the NCO at 7.1 MHz */
    pddc_stage_desc st[3];
    int nst = 1, nt[3] = { 0, 0, 0 };
    float *taps[3] = { NULL, NULL, NULL };
    uint32_t flags = 0, freg = 0;
    if (cascade) {
        static const char *names[3] = { "taps_c320_s1_d8_32.f32", "taps_c320_s2_d8_64.f32", "taps_c320_s3_d5_161.f32" };
        static const int dec[3] = { 8, 8, 5 };
        nst = 3;
        for (int i = 0; i < 3; i++) {
            if (!(taps[i] = load_f32(tapdir, names[i], &nt[i])))
                return 1;
            st[i] = (pddc_stage_desc){ dec[i], nt[i], taps[i], 0 };
        }
        flags = PDDC_F_MIX;
        freg = pddc_nco_freg(7.1e6, PDDC_ADC_CLK_HZ);
    } else {
        if (!(taps[0] = load_f32(tapdir, "taps_d8_127.f32", &nt[0])))
            return 1;
        st[0] = (pddc_stage_desc){ 8, nt[0], taps[0], 0 };
    }

    /* one communicator rank, one pipeline, one input batch and two output buffers per GPU */
    pddc_comm *comm[MAXG] = { 0 };
    pddc_pipeline *pipe[MAXG] = { 0 };
    void *d_in[MAXG] = { 0 }, *d_out[MAXG][2] = { { 0 } }, *d_all = NULL, *arena[MAXG] = { 0 };
    int devs[MAXG];
    for (int g = 0; g < ng; g++)
        devs[g] = g;
    if (gather || ng > 1)
        CHECK(pddc_comm_init_all(comm, ng, devs));
    size_t n_out = 0, cap = 0;
    for (int g = 0; g < ng; g++) {
        CHECK(pddc_set_device(g));
        CHECK(pddc_pipeline_create(&pipe[g], g, st, nst, flags));   /* same host memory: nothing to broadcast */
        /* the cascade's last stage rides along with the next batch's first-stage launch (one launch per step); with the
         * gather each batch's output has to be complete when its transfer is queued, so that leg keeps the in-line chain */
        if (cascade && !gather)
            CHECK(pddc_pipeline_set_overlap(pipe[g], 1));
        CHECK(pddc_pipeline_set_freg(pipe[g], freg));
        cap = pddc_pipeline_max_output(pipe[g], ns) + 8;
        /* Input, inter-stage workspace and the two output buffers of a GPU are cut from ONE arena, at the pair of
         * slots where a read stream and a write stream run fastest against each other (different HBM extent
         * classes; pddc_pipeline_arena_place, include/perseus_ddc.h).  Slot layout: [input | workspace | out 0 | out 1]. */
        const size_t GiB = (size_t)1 << 30, slot = 8 * GiB;
        const size_t in_span = (ns * 6 + GiB - 1) / GiB * GiB;
        const size_t ws = (pddc_pipeline_workspace_size(pipe[g], ns) + 255) & ~(size_t)255;
        const size_t ob = (cap * 8 + 255) & ~(size_t)255;
        size_t got = 0;
        if (arena_gib > 0 && in_span + ws + 2 * ob <= slot) {
            for (size_t gib = (size_t)arena_gib; gib >= 24 && !arena[g]; gib = gib * 3 / 4)
                if (pddc_malloc(&arena[g], gib * GiB) == PDDC_OK)
                    got = gib * GiB;
                else
                    arena[g] = NULL;
        }
        if (arena[g]) {
            size_t si = 0, so = 0;
            float fast = 0, slow = 0;
            /* the pipeline's own first kernel is the probe (input at the arena's start, filled first); a cascade's
             * workspace is set at the chosen slot by the call itself */
            int np = 0;
            CHECK(pddc_synth_lcg(arena[g], ns * 6, 12345u + (uint32_t)g, 0, NULL));
            CHECK(pddc_pipeline_arena_place(pipe[g], arena[g], got, slot, ns, in_span, &so, &slow, &fast, &np, NULL));
            fprintf(stderr, "GPU %d: %zu GiB arena, input at its start, write side in slot %zu after %d probes with the kernel "
                            "itself: %.4f ms (first come %.4f ms)\n", g, got / GiB, so, np, fast, slow);
            d_in[g] = (char *)arena[g] + si * slot;
            char *o = (char *)arena[g] + so * slot + in_span;
            d_out[g][0] = o + ws;
            d_out[g][1] = o + ws + ob;
        } else {
            /* no room for an arena: separate allocations, the outputs walked away from the input */
            CHECK(pddc_malloc(&d_in[g], ns * 6));
            CHECK(pddc_malloc_apart(&d_out[g][0], cap * 8, d_in[g], ns * 6, 24, NULL, NULL));
            CHECK(pddc_malloc_apart(&d_out[g][1], cap * 8, d_in[g], ns * 6, 24, NULL, NULL));
        }
        CHECK(pddc_synth_lcg(d_in[g], ns * 6, 12345u + (uint32_t)g, 0, NULL));     /* stream seed 12345 + g */
        CHECK(pddc_stream_sync(NULL));
    }
    n_out = pddc_pipeline_max_output(pipe[0], ns);
    if (gather) {
        CHECK(pddc_set_device(0));
        CHECK(pddc_malloc(&d_all, (size_t)ng * n_out * 8));
    }

    double t_kernel = 0.0, t_gather = 0.0;
    for (int leg = 0; leg < (gather ? 2 : 1); leg++) {
        double t0 = 0.0;
        for (int k = -warm; k < steps; k++) {
            if (k == 0) {
                for (int g = 0; g < ng; g++) {
                    CHECK(pddc_set_device(g));
                    if (leg == 1)
                        CHECK(pddc_comm_gather_wait(comm[g]));
                    CHECK(pddc_stream_sync(NULL));
                }
                t0 = now_s();
            }
            const int b = k & 1;
            for (int g = 0; g < ng; g++) {                       /* every GPU gets its batch before any is waited for */
                size_t n = 0;
                CHECK(pddc_set_device(g));
                if (leg == 1)
                    CHECK(pddc_comm_gather_fence(comm[g], NULL));  /* the transfer that last read out[b] is done */
                CHECK(pddc_pipeline_process(pipe[g], d_in[g], ns, d_out[g][b], cap, &n, NULL));
            }
            if (leg == 1) {
                CHECK(pddc_comm_group_start());                   /* one collective: a call per communicator */
                for (int g = 0; g < ng; g++)
                    CHECK(pddc_comm_gather_async(comm[g], d_out[g][b], n_out * 8, d_all, 0, NULL));
                CHECK(pddc_comm_group_end());
            }
        }
        for (int g = 0; g < ng; g++) {
            CHECK(pddc_set_device(g));
            if (leg == 1)
                CHECK(pddc_comm_gather_wait(comm[g]));
            CHECK(pddc_pipeline_fence(pipe[g], NULL));          /* overlap mode: the last batch's held-back stage */
            CHECK(pddc_stream_sync(NULL));
        }
        if (leg == 0)
            t_kernel = now_s() - t0;
        else
            t_gather = now_s() - t0;
    }

    const double total = (double)ng * (double)ns * steps;
    printf("%d GPU(s), %s, 2^%d samples per GPU per step, %d steps: %.1f MS/s aggregate (%.4f ms per step)", ng,
           cascade ? "NCO + cascade /320" : "127-tap /8", log2n, steps, total / t_kernel / 1e6, t_kernel / steps * 1e3);
    if (gather)
        printf("; with the gather to GPU 0: %.1f MS/s (%.4f ms per step, %.2f GB/s per peer link, %.2f GB/s into the root)",
               total / t_gather / 1e6, t_gather / steps * 1e3, ng > 1 ? (double)n_out * 8 * steps / t_gather / 1e9 : 0.0,
               (double)(ng - 1) * n_out * 8 * steps / t_gather / 1e9);
    printf("\n");
    /* the same figures as ONE JSON line with bench.py's keys (its N > 1 line: value = the kernels' aggregate, the gather
     * in its own object, never in `value`) so the two hosts can be compared by a script */
    {
        const char *label = cascade ? "80 MS/s synthetic 24-bit I/Q, NCO mix 7.1 MHz + cascade /320 (8*8*5)"
                                    : "80 MS/s synthetic 24-bit I/Q, unpack + 127-tap polyphase decimate-by-8";
        int rccl_run = 0, rccl_hdr = 0;                  /* version codes: the library in the process, the header built against */
        (void)pddc_comm_rccl_version(&rccl_run, &rccl_hdr);
        printf("{\"metric\": \"input MS/s through unpack+decimate, 1/2/4/8 GPU; %% HBM-roofline\", \"value\": %.1f, \"unit\": \"MS/s\", \"n_gpus\": %d, \"steps\": %d, "
               "\"warmup\": %d, \"ms_per_step\": %.4f, \"higher_is_better\": true, \"scaling\": \"weak\", \"vs_baseline\": null, "
               "\"dtype\": \"%s\", \"data\": \"synthetic\", \"host\": \"C (perseus_multi_bench, one process, pddc_comm_init_all)\", "
               "\"config\": {\"workload\": \"%s\", \"samples_per_gpu_per_step\": %zu, \"input\": \"LCG bytes seed 12345+gpu, device "
               "resident\", \"sharding\": \"independent stream per GPU, no data-path collective\"}, \"rccl\": {\"running\": %d, "
               "\"header\": %d}",
               total / t_kernel / 1e6, ng, steps, warm, t_kernel / steps * 1e3,
               /* the arithmetic the first-stage kernel of THIS run computes in (bench.py's rule) */
               pddc_pipeline_stage0_on_i8(pipe[0], ns) > 0 ? "i8xi8->i32, f32 out" : "f32", label, ns, rccl_run, rccl_hdr);
        if (gather)
            printf(", \"gather\": {\"workload\": \"every GPU's output gathered on GPU 0 (pddc_comm_gather_async)\", \"value\": %.1f, "
                   "\"unit\": \"MS/s\", \"ms_per_step\": %.4f, \"out_bytes_per_rank_per_step\": %zu, \"root_ingest_GBps\": %.2f, "
                   "\"per_link_GBps\": %.2f}",
                   total / t_gather / 1e6, t_gather / steps * 1e3, n_out * 8,
                   (double)(ng - 1) * n_out * 8 * steps / t_gather / 1e9,
                   ng > 1 ? (double)n_out * 8 * steps / t_gather / 1e9 : 0.0);
        printf("}\n");
    }

    for (int g = 0; g < ng; g++) {
        pddc_set_device(g);
        pddc_pipeline_destroy(pipe[g]);
        if (arena[g]) {
            pddc_free(arena[g]);
        } else {
            pddc_free(d_in[g]);
            pddc_free(d_out[g][0]);
            pddc_free(d_out[g][1]);
        }
        if (comm[g])
            pddc_comm_destroy(comm[g]);
    }
    if (d_all) {
        pddc_set_device(0);
        pddc_free(d_all);
    }
    for (int i = 0; i < 3; i++)
        free(taps[i]);
    return 0;
}
