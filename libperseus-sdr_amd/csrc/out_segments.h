/*
 * out_segments.h -- where a receiver's decimated output lies between the GPU and the callbacks (perseus_api.c, DDC modes).
 *
 * The reference hands its callbacks library-owned transfer buffers that libusb filled (perseus-in.c:187-264); here the
 * GPU fills ONE pinned buffer and the callbacks read it in place.  Every batch RESERVES its place in that buffer when it
 * is submitted -- `worst` bytes in one piece, behind the newest segment or, when the buffer's end is too near, at its start
 * below the oldest one --, becomes READY once its ticket has been waited for, and is TAKEN from the front buffersize bytes at
 * a time: a pointer into the buffer where those bytes lie in one piece, a gather into the transfer's ring slot where they
 * straddle two segments.  No byte moves on the host otherwise.  Needs every batch to hold at least two buffers' worth
 * (then OSEG_MAX segments always hold a whole buffer and the stream cannot stall); perseus_api.c keeps a byte ring for
 * streams of smaller batches.  Pure C, no GPU: tests/out_segments_test.c drives it against a byte-queue model.
 */
#ifndef PERSEUS_OUT_SEGMENTS_H
#define PERSEUS_OUT_SEGMENTS_H
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define OSEG_MAX 8
#define OSEG_ALIGN 256u

typedef struct {
    int ticket;                 /* pddc staging slot                                  */
    size_t off, len;            /* bytes not yet taken: buf[off .. off + len)          */
} out_seg;

typedef struct {
    out_seg seg[OSEG_MAX];      /* oldest first from `head`; the last n_pend of the n are still being written */
    int head, n, n_pend;
    size_t ready;               /* bytes ready and not yet taken */
    size_t cap;                 /* size of the buffer            */
} out_segs;

static inline size_t oseg_align(size_t x) { return (x + OSEG_ALIGN - 1) & ~(size_t)(OSEG_ALIGN - 1); }

/* where the next batch's output (at most `worst` bytes, in one piece) can go; 0: no room until more has been taken */
static inline int oseg_reserve(const out_segs *s, size_t worst, size_t *off)
{
    size_t at = 0;
    if (s->n == OSEG_MAX)
        return 0;
    if (s->n > 0) {
        const out_seg *h = &s->seg[s->head], *l = &s->seg[(s->head + s->n - 1) % OSEG_MAX];
        const size_t start = h->off, wr = oseg_align(l->off + l->len);
        if (l->off < h->off) {                        /* the newest already lies below the oldest */
            if (wr > start || start - wr < worst)
                return 0;
            at = wr;
        } else if (wr <= s->cap && s->cap - wr >= worst) {
            at = wr;
        } else if (start >= worst) {
            at = 0;
        } else {
            return 0;
        }
    } else if (worst > s->cap) {
        return 0;
    }
    if (off)
        *off = at;
    return 1;
}

static inline void oseg_push(out_segs *s, int ticket, size_t off, size_t len)
{
    s->seg[(s->head + s->n) % OSEG_MAX] = (out_seg){ ticket, off, len };
    s->n++;
    s->n_pend++;
}

/* the oldest segment still being written (n_pend > 0) */
static inline out_seg oseg_oldest_pending(const out_segs *s) { return s->seg[(s->head + s->n - s->n_pend) % OSEG_MAX]; }

/* ... has arrived: its bytes can be taken */
static inline void oseg_ready(out_segs *s)
{
    s->ready += oseg_oldest_pending(s).len;
    s->n_pend--;
}

/* ... or leaves the list without ever becoming ready here (perseus_api.c's ring mode copies it elsewhere; n == n_pend) */
static inline void oseg_pop_pending(out_segs *s)
{
    s->head = (s->head + 1) % OSEG_MAX;
    s->n--;
    s->n_pend--;
}

static inline void oseg_trim(out_segs *s)           /* drop the oldest segments that are ready and empty */
{
    while (s->n > s->n_pend && s->seg[s->head].len == 0) {
        s->head = (s->head + 1) % OSEG_MAX;
        s->n--;
    }
}

/* the next n ready bytes (ready >= n): where they lie in `buf` (*in_place = 1), or gathered into `slot` */
static inline const uint8_t *oseg_take(out_segs *s, const uint8_t *buf, size_t n, uint8_t *slot, int *in_place)
{
    oseg_trim(s);
    out_seg *h = &s->seg[s->head];
    const uint8_t *p = slot;
    if (h->len >= n) {
        p = buf + h->off;
        h->off += n;
        h->len -= n;
        *in_place = 1;
    } else {
        size_t have = 0;
        while (have < n) {                            /* (ready >= n: only ready segments are touched) */
            h = &s->seg[s->head];
            const size_t t = h->len < n - have ? h->len : n - have;
            memcpy(slot + have, buf + h->off, t);
            h->off += t;
            h->len -= t;
            have += t;
            oseg_trim(s);
        }
        *in_place = 0;
    }
    oseg_trim(s);
    s->ready -= n;
    return p;
}
#endif
