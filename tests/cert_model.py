"""Python statement of k_trace_walk's ORDER TEST (csrc/tn_trace_walk.hip: struct OrderR6 -- the part of the certification that
decides whether the reference's dedupe / pairing phases, optix_trace_rays.cu:124-257, reduce to "pair hit k-1 with hit k, drop the
pairs shorter than eps" for a sound chain) -- statement for statement the device code, so that the rules can be checked against the
oracle's LITERAL algorithm on hundreds of thousands of crafted chains without a GPU (tests/test_certification_rules.py).  Test
infrastructure only.

Hits come in CHAIN order.  A hit JOINS the current cluster iff it is less than eps above the cluster's largest t (a new cluster
therefore starts at least eps above every member of the old one).  "Inversion": two adjacent hits whose sorted order (t, face id)
is not chain order.
  * a cluster without an inversion = an ascending run of short gaps: any length;
  * D  a cluster with an inversion: at most THREE members, no member eps or more below an earlier one, every member at least eps
       above the previous cluster (two members: the isolated inverted pair of rounds 2-5);
  * A  the LAST cluster may be an inverted PAIR when the chain has >= 4 hits (with 3 hits the entry hull face would look ahead straight
       at the exit hull face: get_common_tetrahedra's EMPTY == EMPTY);
  * B  an ascending FIRST cluster of >= 3 hits: phase 1 clears its interior, the entry face's look-ahead examines the cluster's last
       face and the next one and stops iff the gap behind that one is long -- required: two more cluster starts must follow, each
       a single hit when the next starts (`pend`);
  * C  the first cluster = an inverted pair: after the sort hit 1 pairs with hit 0 (short: nothing emitted) and hit 0 finds no
       partner, so the reference LOSES the segment of hit 2 (drop2); three single-hit clusters must follow.
`pend`: 0 none | 1 inside B's run | 2 one more long gap needed | 3 C: hit 2 pending | 4 two more long gaps needed; a hit that joins
a cluster while pend >= 2 ends the certification.
"""
import numpy as np

EPS = np.float32(1e-6)
PEND_NEW_CLUSTER = (0, 2, 0, 4, 2)


def certify(t, fid):
    """t: float32 [n+1] in chain order, fid: face ids.  Returns (certified, drop2, rules used)."""
    ok = True
    cinv = cfirst = d2 = False
    cn = pend = 0
    cmax = prev_cmax = np.float32(0)
    pt = np.float32(0)
    have_prev = False
    nhits = 0
    used = {"A": False, "B": False, "C": False, "D": False}
    for k in range(len(t)):
        ct = np.float32(t[k])
        vp, start = have_prev, not have_prev
        joins = not (np.float32(ct - cmax) >= EPS)
        asc = bool(ct > pt or (ct == pt and k > 0 and fid[k] > fid[k - 1]))
        vj, vn = vp and joins, vp and not joins
        long_inv = np.float32(cmax - ct) >= EPS
        cinv_n = cinv or not asc
        cn_n = cn + 1
        inv_ok = (not long_inv) and ((cn_n == 2) if cfirst else (cn_n <= 3 and np.float32(ct - prev_cmax) >= EPS))
        good = pend < 2 and ((not cinv_n) or inv_ok)
        ok = ok and ((not vj) or good)
        rule_c = vj and cinv_n and cfirst and cn_n == 2
        rule_b = vj and (not cinv_n) and cfirst and cn_n == 3
        d2 = d2 or rule_c
        used["B"] |= bool(rule_b); used["C"] |= bool(rule_c); used["D"] |= bool(vj and cinv_n and not cfirst and cn_n == 3 and good)
        pend = (3 if rule_c else (1 if rule_b else pend)) if vj else (PEND_NEW_CLUSTER[pend] if vn else pend)
        fresh = vn or start
        prev_cmax = cmax if vn else prev_cmax
        cmax = max(cmax, ct) if vj else (ct if fresh else cmax)
        cn = cn_n if vj else (1 if fresh else cn)
        cinv = cinv_n if vj else (False if fresh else cinv)
        cfirst = True if start else (False if vn else cfirst)
        pt = ct
        have_prev = True
        nhits += 1
    fin = ok and pend == 0 and ((not cinv) or ((not cfirst) and cn == 2 and nhits >= 4))
    used["A"] = bool(fin and cinv)
    return bool(fin), bool(d2), {k: bool(v and fin) for k, v in used.items()}


PEND_NEXT_ON_LONG = PEND_NEW_CLUSTER


def certify_pairwise(t, fid):
    """struct OrderPair<true> (OrderR5e): round 5's pairwise test + rules A-C -- what the tracer runs below 500k tets.  Pairwise
    notions: "short" = |t_k - t_(k-1)| < eps; an inverted pair must be isolated (the pair before long, the face before it at least
    eps away, a long gap clear of both members behind it); no rule D."""
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
    return bool(order_ok), bool(drop2), {"A": bool(order_ok and prev_inv), "B": bool(order_ok and used_b), "C": bool(order_ok and drop2),
                                         "D": False}


def plain_pairing(t, drop2):
    """What the segment writer emits for a certified chain: (tet index k, t_in, t_out) for every pair (k-1, k) that is not short."""
    segs = []
    for k in range(1, len(t)):
        if abs(np.float32(t[k - 1] - t[k])) < EPS or (drop2 and k == 2):
            continue
        segs.append((k - 1, float(t[k - 1]), float(t[k])))
    return segs
