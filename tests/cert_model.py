"""Python statement of k_trace_walk's ORDER TEST (csrc/tn_trace_walk.hip: the part of the certification that decides whether the
reference's dedupe / pairing phases, optix_trace_rays.cu:124-257, reduce to "pair hit k-1 with hit k, drop the pairs shorter than
eps" for a sound chain) -- line for line the device code's rules, so that they can be checked against the oracle's LITERAL
algorithm on millions of crafted chains without a GPU (tests/test_certification_rules.py).  Test infrastructure only.

Rules (hits in CHAIN order; "short" = |t_k - t_(k-1)| < eps; "asc" = sorted order of the pair equals chain order, exact ties by
face id):
  long asc                     fine (after an inverted pair: only when clear of both of its members by eps)
  short asc                    fine unless the pair before was inverted
  short inverted               fine when isolated: the pair before is long and the face before it at least eps away
  round 6, A  an isolated inverted pair at the very END of a chain of >= 4 hits (no following face is needed to clear it; with 3
              hits the entry hull face would look ahead straight at the exit hull face: get_common_tetrahedra's EMPTY == EMPTY)
  round 6, B  a run of >= 2 short ascending gaps AT THE ENTRY face: phase 1 clears the run's interior, the entry face's look-ahead
              examines the run's last face and the next one and stops iff the gap behind that one is long -- required (two long
              ascending gaps after the run), because a look-ahead that reaches the exit hull face pairs the two hull faces
  round 6, C  the FIRST pair inverted by less than eps: after the sort hit 1 pairs with hit 0 (short: nothing emitted) and hit 0
              finds no partner, so the reference LOSES the segment of hit 2 (drop2); hit 2 must be clear of both by eps and two
              more long ascending gaps must follow (the same look-ahead bound)
"""
import numpy as np

EPS = np.float32(1e-6)
PEND_NEXT_ON_LONG = (0, 2, 0, 4, 2)     # pend: 0 none | 1 inside the entry run | 2 one more long gap needed | 3 first pair inverted: hit 2
                                        # pending | 4 two more long gaps needed


def certify(t, fid):
    """t: float32 [n+1] in chain order, fid: face ids.  Returns (certified, drop2, rules used)."""
    have_prev = have_pp = False
    order_ok = True
    prev_short = prev_inv = False
    pt = ppt = np.float32(0)
    nhits = 0
    pend = 0
    drop2 = used_b = False
    for k in range(len(t)):
        ct = np.float32(t[k])
        if have_prev:
            is_short = abs(np.float32(pt - ct)) < EPS
            asc = bool(ct > pt or (ct == pt and fid[k] > fid[k - 1]))
            clear2 = np.float32(ct - ppt) >= EPS
            pend_wait = pend >= 2
            first_inv = is_short and not asc and not have_pp                      # rule C
            entry_run = is_short and asc and prev_short and nhits == 2            # rule B
            if is_short:
                ok = (not pend_wait) and ((not prev_inv) if asc else ((not have_pp) or (not prev_short and clear2)))
                pend = 1 if entry_run else (3 if first_inv else pend)
            else:
                ok = asc and ((not prev_inv) or clear2)
                pend = PEND_NEXT_ON_LONG[pend]
            drop2 = drop2 or first_inv
            used_b = used_b or entry_run
            order_ok = order_ok and ok
            prev_inv = is_short and not asc
            prev_short = is_short
        nhits += 1
        have_pp = have_prev
        ppt = pt
        pt = ct
        have_prev = True
    order_ok = order_ok and pend == 0 and not (prev_inv and nhits < 4)            # rule A (+ B / C settled before the chain ends)
    return bool(order_ok), bool(drop2), {"A": bool(order_ok and prev_inv), "B": bool(order_ok and used_b), "C": bool(order_ok and drop2)}


def plain_pairing(t, drop2):
    """What the segment writer emits for a certified chain: (tet index k, t_in, t_out) for every pair (k-1, k) that is not short."""
    segs = []
    for k in range(1, len(t)):
        if abs(np.float32(t[k - 1] - t[k])) < EPS or (drop2 and k == 2):
            continue
        segs.append((k - 1, float(t[k - 1]), float(t[k])))
    return segs
