/* tests/out_segments_test.c -- the segment allocator of the drop-in API's delivery path (csrc/out_segments.h) against a byte
 * queue: random batch lengths (each at least two buffers' worth, at most `worst`), random interleaving of submit / arrive /
 * deliver, the "GPU" writing a running byte counter into every reservation.  Checked: a reservation never touches bytes that
 * are still to be delivered or still being written; the delivered stream is the counter, in order, whether a buffer came
 * in place or gathered; the stream never stalls (with nothing in flight and less than a buffer ready, a reservation must be
 * possible); in place is the rule, gathering the exception (at most one buffer per batch).
 * usage: out_segments_test <seed> <steps> <bufsize> <worst>   -> prints a summary line, exit 0 on success */
#include <stdio.h>
#include <stdlib.h>
#include "out_segments.h"

static uint32_t rng_state;
static uint32_t rnd(void) { return rng_state = rng_state * 1664525u + 1013904223u; }

int main(int argc, char **argv)
{
    if (argc < 5)
        return 2;
    rng_state = (uint32_t)atoi(argv[1]);
    const long steps = atol(argv[2]);
    const size_t bufsize = (size_t)atol(argv[3]), worst = (size_t)atol(argv[4]);
    out_segs s;
    memset(&s, 0, sizeof(s));
    s.cap = 6 * oseg_align(worst) + oseg_align(2 * bufsize);          /* perseus_api.c's sizing */
    uint8_t *buf = malloc(s.cap), *live = calloc(s.cap, 1), *slot = malloc(bufsize);
    uint64_t produced = 0, consumed = 0;       /* the byte counter: byte i of the stream is (uint8_t)(i * 7 + (i >> 8)) */
    long batches = 0, in_place = 0, gathered = 0, stalls = 0;
    for (long it = 0; it < steps; ++it) {
        const uint32_t r = rnd() >> 8;
        const int can_deliver = s.ready >= bufsize;
        const int may_submit = s.n_pend < 2 && s.n - s.n_pend <= 2;   /* perseus_api.c's can_submit */
        size_t off = 0;
        const int can_reserve = may_submit && oseg_reserve(&s, worst, &off);
        if (!can_deliver && s.n_pend == 0 && !can_reserve) {
            printf("STALL at step %ld: ready %zu, segments %d\n", it, s.ready, s.n);
            return 1;
        }
        const int what = r % 3;
        if (what == 0 && can_reserve) {
            /* submit: the GPU will write `len` bytes at off .. (it does so at once here) */
            size_t len = 2 * bufsize + 8 * ((rnd() >> 8) % ((worst - 2 * bufsize) / 8 + 1));
            if ((rnd() & 31) == 0)
                len = worst;
            for (size_t i = 0; i < worst; ++i)
                if (live[off + i]) {
                    printf("reservation [%zu, +%zu) overlaps live byte %zu at step %ld\n", off, worst, off + i, it);
                    return 1;
                }
            for (size_t i = 0; i < len; ++i) {
                const uint64_t k = produced + i;
                buf[off + i] = (uint8_t)(k * 7 + (k >> 8));
                live[off + i] = 1;
            }
            produced += len;
            oseg_push(&s, (int)(batches & 1), off, len);
            batches++;
        } else if (what == 1 && s.n_pend > 0) {
            oseg_ready(&s);                     /* the oldest batch in flight has arrived */
        } else if (can_deliver) {
            int ip = 0;
            const uint8_t *p = oseg_take(&s, buf, bufsize, slot, &ip);
            for (size_t i = 0; i < bufsize; ++i) {
                const uint64_t k = consumed + i;
                if (p[i] != (uint8_t)(k * 7 + (k >> 8))) {
                    printf("byte %llu of the stream is wrong (step %ld, %s)\n", (unsigned long long)k, it, ip ? "in place" : "gathered");
                    return 1;
                }
            }
            consumed += bufsize;
            if (ip)
                in_place++;
            else
                gathered++;
            /* what has been delivered is free again: mark by stream position -> clear `live` through the segment list */
            memset(live, 0, s.cap);
            for (int k = 0; k < s.n; ++k) {
                const out_seg *g = &s.seg[(s.head + k) % OSEG_MAX];
                memset(live + g->off, 1, g->len);
            }
        } else {
            stalls++;                           /* nothing to do this step (e.g. waiting for an arrival) */
        }
    }
    if (gathered > batches) {
        printf("more gathered buffers (%ld) than batches (%ld)\n", gathered, batches);
        return 1;
    }
    printf("ok: %ld batches, %llu bytes delivered, %ld buffers in place, %ld gathered, %ld idle steps\n", batches,
           (unsigned long long)consumed, in_place, gathered, stalls);
    free(buf);
    free(live);
    free(slot);
    return 0;
}
