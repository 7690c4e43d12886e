"""Integer model of the exact time-parallel resampler schedule (see pg_source_dev.h: sched_parallel) next to the serial f32
recurrence of the reference (cubic.rs:72-90). `parallel` returns None where the device code would restart the closed form."""
import numpy as np
rng = np.random.default_rng(1)
TWO24 = 1 << 24
def serial(sp0, ratio, n):
    sp = np.float32(sp0); ratio = np.float32(ratio); one = np.float32(1.0)
    cc = 0; oc = np.zeros(n, np.int64); of = np.zeros(n, np.float32)
    for k in range(n):
        ge = sp >= one
        cc += int(ge)
        if ge: sp = np.float32(sp - one)
        oc[k] = cc; of[k] = sp
        sp = np.float32(sp + ratio)
    return oc, of, sp
def rho(A):
    return np.where((A >= TWO24) & (A & 1 == 1), np.where((A & 3) == 3, 1, -1), 0)
def parallel(sp0, ratio, n):
    S0 = float(np.float32(sp0)) * TWO24; R = float(np.float32(ratio)) * TWO24
    if S0 != int(S0) or R != int(R) or not (0 <= S0 < 2 * TWO24) or not (TWO24 // 2 <= R < TWO24): return None
    S0 = int(S0); R = int(R)
    k = np.arange(n + 1, dtype=np.int64)
    V = S0 + k * R
    P0 = V % TWO24                       # post-wrap, unperturbed, step k
    U0 = np.empty(n + 1, np.int64); U0[0] = S0; U0[1:] = P0[:-1] + R   # pre-wrap, unperturbed
    side0 = U0 >= TWO24
    # tables T_k[m], k = 1..n : delta_k = delta_{k-1} + T_k[delta_{k-1} & 3]
    T = np.zeros((n + 1, 4), np.int64)
    for m in range(4):
        A = U0 + m
        T[:, m] = np.where(side0 & ((A & 1) == 1), np.where((A & 3) == 3, 1, -1), 0)
    T[0, :] = 0
    # scan (sequential composition here; on the GPU a Kogge-Stone over 4-entry tables): delta_k
    d = 0; delta = np.zeros(n + 1, np.int64)
    for kk in range(1, n + 1):
        d = d + T[kk, d & 3]; delta[kk] = d
    U = U0 + delta
    A = U0.copy(); A[1:] = U0[1:] + delta[:-1]
    ok = np.all((U >= TWO24) == side0) and np.all((A >= TWO24) == side0) and np.all(U < 2 * TWO24) and np.all(U >= 0)
    if not ok: return None
    P = U - TWO24 * side0
    cc = np.cumsum(side0[:n].astype(np.int64))
    return cc, (P[:n].astype(np.float64) / TWO24).astype(np.float32), np.float32(U[n] / TWO24)


# ---- ratio in [1, 2): cubic.rs:92-111 (pg_source_dev.h: sched_parallel_up) --------------------------------------------------------
# Per output frame:  while sub_pos < ratio { push an input frame; sub_pos += 1 };  sub_pos -= ratio;  emit (consumed, 1 - sub_pos).
# In units of 2^-23 (the ulp of ratio and of every value in [1, 2)): the loop-top state S is an integer in [0, ONE], the first push is exact,
# a second push lands in [2, 3) where the ulp is two units — an odd value is a tie and rounds to the multiple of four — and the subtraction is exact.
ONE = 1 << 23
def serial_up(sp0, ratio, n):
    sp = np.float32(sp0); ratio = np.float32(ratio); one = np.float32(1.0)
    cc = 0; oc = np.zeros(n, np.int64); of = np.zeros(n, np.float32)
    for k in range(n):
        while sp < ratio:
            cc += 1
            sp = np.float32(sp + one)
        sp = np.float32(sp - ratio)
        oc[k] = cc; of[k] = np.float32(one - sp)
    return oc, of, sp
def parallel_up(sp0, ratio, n):
    """ratio in [1, 2): unit 2^-23, K = 2 pushes at most, the second one rounds. ratio in [2, 4): unit 2^-22, K = 3 or 4; only a fourth push rounds."""
    SH = 23 if np.float32(ratio) < 2 else 22
    ONE = 1 << SH
    S0 = float(np.float32(sp0)) * ONE; R = float(np.float32(ratio)) * ONE
    if S0 != int(S0) or R != int(R) or not (0 <= S0 < ONE) or not (ONE <= R < 4 * ONE): return None
    S0 = int(S0); R = int(R); K = R // ONE + 1; D = K * ONE - R
    rounds = K != 3
    j = np.arange(n + 1, dtype=np.int64)
    X0 = (S0 + j * D) % ONE                 # unrounded loop-top state of output j
    nowrap0 = X0 + D < ONE                  # K pushes (else K - 1)
    T = np.zeros((n + 1, 4), np.int64)      # step j: delta_{j+1} = delta_j + T_j[delta_j & 3]
    for m in range(4):
        x = (X0 + m) & 3
        T[:, m] = np.where(nowrap0 & ((x & 1) == 1) & rounds, x - 2, 0)
    d = 0; delta = np.zeros(n + 2, np.int64)
    for kk in range(n + 1):
        d = d + T[kk, d & 3]; delta[kk + 1] = d
    S = X0 + delta[:n + 1]                  # true loop-top states while the decisions agree
    ok = np.all((S[:n] + D < ONE) == nowrap0[:n]) and np.all(S >= 0) and np.all(S <= ONE)
    if not ok: return None
    wraps = (S0 + (j[1:]) * D) // ONE       # wraps among the first j+1 steps
    cc = K * j[1:] - wraps
    return cc[:n], ((ONE - S[1:n + 1]).astype(np.float64) / ONE).astype(np.float32), np.float32(S[n] / ONE)
