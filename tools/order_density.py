#!/usr/bin/env python3
"""Minimizer runs per k-mer (k = 31, 16-mer minimizers, canonical) of candidate ORDERS on uniform text: round 3s open / closed
syncmer classes over mm27 against round 4s one-hash-per-position form (class + xor / sum of the eleven 6-mer hashes). numpy only."""
import numpy as np, sys
rng = np.random.default_rng(7)
n = 3_000_000
seq = rng.integers(0, 4, n, dtype=np.uint8)
K = 31; W = K - 15
def mers(seq, m):
    # left-aligned in m*2 bits as integer value
    v = np.zeros(len(seq) - m + 1, dtype=np.uint64)
    for i in range(m):
        v = (v << np.uint64(2)) | seq[i:len(seq) - m + 1 + i].astype(np.uint64)
    return v
def rc_of(v, m):
    r = np.zeros_like(v)
    x = v.copy()
    for i in range(m):
        r = (r << np.uint64(2)) | (np.uint64(3) - (x & np.uint64(3)))
        x >>= np.uint64(2)
    return r
m16 = mers(seq, 16); r16 = rc_of(m16, 16); cm = np.minimum(m16, r16).astype(np.uint64)
m6 = mers(seq, 6); r6 = rc_of(m6, 6); c6 = np.minimum(m6, r6)
M32 = np.uint64(0xFFFFFFFF)
def mm27(cm):
    u = (cm * np.uint64(0x9E3779B1)) & M32
    u ^= u >> np.uint64(15)
    u &= np.uint64(0xFFFFFFE0)
    return (u * np.uint64(0x85EBCA6B)) & M32
def density(key):
    # sliding argmin over W (leftmost on ties); runs per k-mer
    nk = len(key) - W + 1
    idx = np.arange(nk)
    best = key[:nk].copy(); arg = idx.copy()
    for j in range(1, W):
        kj = key[j:j + nk]
        b = kj < best
        best = np.where(b, kj, best); arg = np.where(b, idx + j, arg)
    runs = 1 + np.count_nonzero(arg[1:] != arg[:-1])
    return runs / nk
n16 = len(cm)
def cls_old(u):
    # u: hashes per 6-mer position; class of 16-mer at p
    win = np.stack([u[i:i + n16] for i in range(11)])
    sides = np.minimum(win[:5].min(0), win[6:].min(0))
    open_ = win[5] < sides
    ends = np.minimum(win[0], win[10]); inner = win[1:10].min(0)
    return np.where(open_, 0, np.where(ends <= inner, 1, 2)).astype(np.uint64), win
def cls_new(u):
    win = np.stack([u[i:i + n16] for i in range(11)])
    ends = np.minimum(win[0], win[10]); inner = win[1:10].min(0)
    return np.where(ends <= inner, 1, np.where(win[5] == inner, 0, 2)).astype(np.uint64), win
u_old = (c6 * np.uint64(0x9E3779B1)) & M32
print("random mm27      ", density(mm27(cm)))
c, _ = cls_old(u_old)
print("old OC + mm27    ", density((c << np.uint64(30)) | ((mm27(cm) >> np.uint64(2)) & np.uint64(0xFFFFFFE0))))
for name, A in (("A24=0x9E3779", 0x9E3779), ("A32=0x9E3779B1", 0x9E3779B1), ("A24=0x85EBCB", 0x85EBCB)):
    u = (c6 * np.uint64(A)) & M32
    c, win = cls_new(u)
    hx = win[0].copy()
    for i in range(1, 11): hx ^= win[i]
    hs = win.sum(0) & M32
    kx = (c << np.uint64(30)) | (hx & np.uint64(0x3FFFFFE0))
    ks = (c << np.uint64(30)) | (hs & np.uint64(0x3FFFFFE0))
    km = (c << np.uint64(30)) | ((mm27(cm) >> np.uint64(2)) & np.uint64(0xFFFFFFE0))
    print(name, "new cls + mm27", density(km), " xor11", density(kx), " sum11", density(ks))
    # ties: fraction of windows whose min key is attained by >1 distinct position
    nk = len(kx) - W + 1
    best = np.stack([kx[j:j + nk] for j in range(W)]).min(0)
    cnt = sum((kx[j:j + nk] == best) for j in range(W))
    print("   windows with tied minimum (xor11):", np.mean(cnt > 1))
    best = np.stack([km[j:j + nk] for j in range(W)]).min(0)
    cnt = sum((km[j:j + nk] == best) for j in range(W))
    print("   windows with tied minimum (mm27):", np.mean(cnt > 1))
