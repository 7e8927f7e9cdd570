"""GPU tests (-m gpu) of gang submission (pddc_gang_*, include/perseus_ddc.h): several pipelines that share a GPU --
the drop-in API's virtual receivers, the reference's eight descriptors behind one poll thread (perseus-sdr.c:43-47,
736-774) -- push their batches through ONE launch chain, the receiver being the grid's second dimension.  The kernels'
code and every receiver's arguments are those of a push of its own, so the outputs must be the SAME BITS as the
per-pipeline path (and, like it, within 1e-6 of full scale of the CPU oracle)."""
import numpy as np
import pytest

from conftest import load_taps

pytestmark = pytest.mark.gpu
FIR_TOL = 1e-6
FREG = 381178347


def lowpass(ntaps, cutoff):
    k = np.arange(ntaps) - (ntaps - 1) / 2.0
    h = np.sinc(2 * cutoff * k) * np.hamming(ntaps)
    return (h / h.sum()).astype(np.float32)


def plans():
    h1, h2, h3 = load_taps("c320_s1_d8_32"), load_taps("c320_s2_d8_64"), load_taps("c320_s3_d5_161")
    return {
        "8*8*5": [(8, h1), (8, h2), (5, h3)],                               # 250 kS/s (BASELINE config 3)
        "8*8*10": [(8, lowpass(48, 0.05)), (8, lowpass(51, 0.05)), (10, lowpass(287, 0.04))],      # 125 kS/s
        "8*10": [(8, lowpass(64, 0.05)), (10, lowpass(161, 0.04))],         # 1 MS/s
        "8*5": [(8, load_taps("d8_127")), (5, lowpass(81, 0.08))],          # 2 MS/s shape, 127-tap first stage
        "8*8": [(8, h1), (8, h2)],                                          # the pair alone
        "8": [(8, load_taps("d8_255"))],                                    # the first stage alone (255 taps)
        "8*7": [(8, h1), (7, lowpass(57, 0.06))],                           # tail on the generic decimator
        "10*5": [(10, lowpass(97, 0.04)), (5, lowpass(81, 0.08))],          # not gang-able: first stage is not /8
        "10*5m": [(10, lowpass(51, 0.04)), (5, lowpass(117, 0.08))],        # 1.6 MS/s: /10 on the matrix cores (k_fir_i8x<.., 10>), a chain of its own
        "8*8*4*5": [(8, h1), (8, h2), (4, lowpass(33, 0.1)), (5, lowpass(41, 0.08))],   # not gang-able: two stages behind
    }


class Rx:
    """one receiver: a pipeline, a seed, pinned output slots"""

    def __init__(self, pkg, stages, seed, nmax, mix=True, freg=FREG):
        self.pipe = pkg.Pipeline(stages, mix=mix)
        if mix:
            self.pipe.set_freg(freg)
        self.seed = seed
        self.cap = self.pipe.max_output(nmax) + 8
        self.bufs = [pkg.PinnedBuffer(self.cap * 8) for _ in range(2)]
        self.pos = 0
        self.out = []

    def item(self, slot_hint, h_packed=None):
        d = {"pipe": self.pipe, "h_out": self.bufs[slot_hint].ptr, "out_cap": self.cap}
        if h_packed is not None:
            d["h_packed"] = h_packed
        else:
            d["seed"], d["byte_offset"] = self.seed, 6 * self.pos
        return d

    def take(self, slot, n_out):
        self.out.append(self.bufs[slot].array[:8 * n_out].view(np.float32).copy())

    def close(self):
        self.pipe.close()
        for b in self.bufs:
            b.free()


def run_solo(pkg, stages, seed, sizes, nmax, mix=True, retune=None):
    r = Rx(pkg, stages, seed, nmax, mix)
    for k, ns in enumerate(sizes):
        if retune and k in retune:
            r.pipe.set_freg(retune[k])
        n_out, t = r.pipe.push_synth_async(seed, 6 * r.pos, ns, r.bufs[k & 1].ptr, r.cap)
        r.pipe.wait_ticket(t)
        r.take(k & 1, n_out)
        r.pos += ns
    y = np.concatenate(r.out)
    r.close()
    return y


def run_gang(pkg, plan_list, seeds, sizes, nmax, mix=True, retune=None, expect_ganged=None):
    gang = pkg.Gang(0)
    rx = [Rx(pkg, st, sd, nmax, mix) for st, sd in zip(plan_list, seeds)]
    shared = []
    for k, ns in enumerate(sizes):
        if retune and k in retune:
            rx[1].pipe.set_freg(retune[k])
        res, ng = gang.push_async([r.item(k & 1) for r in rx], ns)
        shared.append(ng)
        for r, (n_out, t) in zip(rx, res):
            r.pipe.wait_ticket(t)
            r.take(k & 1, n_out)
            r.pos += ns
    ys = [np.concatenate(r.out) for r in rx]
    for r in rx:
        r.close()
    gang.close()
    if expect_ganged is not None:
        assert shared == expect_ganged, shared
    return ys


SIZES = [1 << 16, 3 << 14, 1 << 18, 4096, 5 << 12, 1 << 17]      # whole tiles (the pair's route), uneven


@pytest.mark.parametrize("plan", ["8*8*5", "8*8*10", "8*10", "8*5", "8*8", "8", "8*7"])
@pytest.mark.parametrize("n", [2, 8])
def test_gang_round_is_bit_identical_to_pushes_of_its_own(pkg, O, dev, plan, n):
    stages = plans()[plan]
    seeds = [777 + 13 * i for i in range(n)]
    ys = run_gang(pkg, [stages] * n, seeds, SIZES, max(SIZES), expect_ganged=[n] * len(SIZES))
    for i in (0, n - 1):
        solo = run_solo(pkg, stages, seeds[i], SIZES, max(SIZES))
        assert ys[i].size == solo.size and np.array_equal(ys[i].view(np.uint32), solo.view(np.uint32)), (plan, n, i)
    assert len({y.tobytes() for y in ys}) == n
    ns = sum(SIZES)
    ref = O.ddc_chain(O.lcg_bytes(6 * ns, seeds[0]), stages, freg=FREG, mix=True)
    assert O.rel_err(ys[0], ref) <= FIR_TOL


def test_gang_with_different_plans_retunes_and_members_that_cannot_share(pkg, O, dev):
    """Eight receivers at six rates in one round: the ones with the same kernels share a launch, the three whose plan is
    not a /8 first stage + one decimator run as chains of their own (one of them with its /10 first stage on the matrix cores) on the gang's stream; receiver 1 is retuned twice
    while streaming (the new word at the batch boundary, phase-continuous).  Every stream equals its solo run."""
    P = plans()
    names = ["8*8*5", "8*8*5", "8*10", "10*5", "8*8*5", "10*5m", "8*8*4*5", "8*8*10"]
    seeds = [31 + i for i in range(8)]
    retune = {2: 123456789, 4: 3000000000}
    ys = run_gang(pkg, [P[k] for k in names], seeds, SIZES, max(SIZES), retune=retune, expect_ganged=[5] * len(SIZES))
    for i, k in enumerate(names):
        solo = run_solo(pkg, P[k], seeds[i], SIZES, max(SIZES), retune=retune if i == 1 else None)
        assert np.array_equal(ys[i].view(np.uint32), solo.view(np.uint32)), (i, k)


def test_gang_sizes_that_leave_the_fused_pair_and_small_batches(pkg, O, dev):
    """Batches that are not whole tiles take the unfused first stage (kind 1 + generic route behind it is not one
    decimator for a three-stage plan: the member runs alone that round), batches shorter than the history likewise;
    the stream state is shared between the routes, so the result must not notice."""
    stages = plans()["8*8*5"]
    sizes = [1 << 16, 4096 + 64, 1 << 15, 64, 128, 1 << 16, 8192 + 8, 1 << 14]
    seeds = [5, 6, 7]
    ys = run_gang(pkg, [stages] * 3, seeds, sizes, max(sizes))
    for i in range(3):
        solo = run_solo(pkg, stages, seeds[i], sizes, max(sizes))
        assert np.array_equal(ys[i].view(np.uint32), solo.view(np.uint32)), i


def test_gang_host_fed_members_and_change_between_gang_and_solo(pkg, O, dev):
    """Host batches (one H2D copy per member) next to on-device sources in one round, and a pipeline that leaves the
    gang for pushes of its own and comes back: still the same stream."""
    stages = plans()["8*8*10"]
    nb, ns = 6, 1 << 16
    host = O.lcg_bytes(6 * ns * nb, 99)
    hin = [pkg.PinnedBuffer(6 * ns) for _ in range(2)]
    gang = pkg.Gang(0)
    rx = [Rx(pkg, stages, 99, ns), Rx(pkg, stages, 100, ns), Rx(pkg, stages, 101, ns)]
    for k in range(nb):
        hin[k & 1].array[:] = host[6 * ns * k:6 * ns * (k + 1)]
        if k in (2, 3):                      # receiver 2 by itself for two batches, the others as a gang of two
            n_out, t = rx[2].pipe.push_synth_async(101, 6 * rx[2].pos, ns, rx[2].bufs[k & 1].ptr, rx[2].cap)
            members = rx[:2]
        else:
            members = rx
        items = [r.item(k & 1, h_packed=hin[k & 1].ptr if r is rx[0] else None) for r in members]
        res, ng = gang.push_async(items, ns)
        assert ng == len(members)
        if k in (2, 3):
            rx[2].pipe.wait_ticket(t)
            rx[2].take(k & 1, n_out)
            rx[2].pos += ns
        for r, (n_out, t) in zip(members, res):
            r.pipe.wait_ticket(t)
            r.take(k & 1, n_out)
            r.pos += ns
    for i, r in enumerate(rx):
        y = np.concatenate(r.out)
        solo = run_solo(pkg, stages, 99 + i, [ns] * nb, ns)
        assert np.array_equal(y.view(np.uint32), solo.view(np.uint32)), i
        r.close()
    gang.close()
    for b in hin:
        b.free()


def test_gang_refuses_bad_rounds_without_moving_any_stream(pkg, O, dev):
    stages = plans()["8*8*5"]
    gang = pkg.Gang(0)
    a, b = Rx(pkg, stages, 1, 1 << 16), Rx(pkg, stages, 2, 1 << 16)
    small = b.item(0)
    small["out_cap"] = 3
    with pytest.raises(pkg.PddcError):
        gang.push_async([a.item(0), small], 1 << 16)          # the second member's buffer is too small
    with pytest.raises(pkg.PddcError):
        gang.push_async([a.item(0), a.item(1)], 1 << 16)      # the same pipeline twice
    with pytest.raises(pkg.PddcError):
        gang.push_async([a.item(0), b.item(0)], 1001)         # not a multiple of the input granule
    res, ng = gang.push_async([a.item(0), b.item(0)], 1 << 16)
    for r, (n_out, t) in zip((a, b), res):
        r.pipe.wait_ticket(t)
        r.take(0, n_out)
    solo = run_solo(pkg, stages, 1, [1 << 16], 1 << 16)
    assert np.array_equal(a.out[0].view(np.uint32), solo.view(np.uint32))      # refused rounds left no trace
    a.close()
    b.close()
    gang.close()


def test_gang_members_in_overlap_mode_run_chains_of_their_own(pkg, O, dev):
    """A pipeline in overlap mode holds its last stage back for its own next launch: such a member does not share the
    round's kernels (its batch runs on the gang's stream as a chain of its own, fenced), and its stream is the same."""
    stages = plans()["8*8*5"]
    gang = pkg.Gang(0)
    rx = [Rx(pkg, stages, 11 + i, max(SIZES)) for i in range(3)]
    rx[1].pipe.set_overlap(True)
    shared = []
    for k, ns in enumerate(SIZES):
        res, ng = gang.push_async([r.item(k & 1) for r in rx], ns)
        shared.append(ng)
        for r, (n_out, t) in zip(rx, res):
            r.pipe.wait_ticket(t)
            r.take(k & 1, n_out)
            r.pos += ns
    assert shared == [2] * len(SIZES)
    for i, r in enumerate(rx):
        solo = run_solo(pkg, stages, 11 + i, SIZES, max(SIZES))
        assert np.array_equal(np.concatenate(r.out).view(np.uint32), solo.view(np.uint32)), i
        r.close()
    gang.close()


@pytest.mark.parametrize("name", ["d8_127", "d8_255"])
def test_untuned_long_first_stages_keep_their_bits_in_a_gang(pkg, O, dev, name):
    """Without the NCO a long first stage runs on k_fir_i8x's plain form, in a gang as one k_fir_i8x_many launch for all its
    members (until round 5: round 3's k_fir_i8, which had no many-stream launch -- such members ran chains of their own):
    the bits of a push of its own either way, which the vector kernel of a shared launch would not give (1e-7 apart)."""
    stages = [(8, load_taps(name))]
    seeds = [5, 6, 7]
    ys = run_gang(pkg, [stages] * 3, seeds, SIZES, max(SIZES), mix=False)
    for i in range(3):
        solo = run_solo(pkg, stages, seeds[i], SIZES, max(SIZES), mix=False)
        assert np.array_equal(ys[i].view(np.uint32), solo.view(np.uint32)), i
    ref = O.ddc_chain(O.lcg_bytes(6 * sum(SIZES), seeds[0]), stages)
    assert O.rel_err(ys[0], ref) <= FIR_TOL
