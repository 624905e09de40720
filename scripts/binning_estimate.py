#!/usr/bin/env python3
"""What would workload-binned placement buy?  (round-2 review, item 1a: estimate before building.)

A measurement with the SHIPPED kernel instead of a model: a steady-state batch is played forward, and at sample ticks the very
same records are stepped ONE tick in several placements — the envs of the device buffer are permuted (whole columns of the
struct-of-arrays state; envs never interact, the move stream is keyed by the env's index and i.i.d., so a permutation changes
nothing statistically) so that envs of the same next-tick class share 16-env wavefronts:

  X  the head of the bomb queue goes off in the next tick (timer 1 now)        -> top_explosions / explode
  F  the head of the flame queue runs out in the next tick (timeLeft 1 now)    -> flame_pops
  R  the env is finished and restarts at the start of the next tick            -> restart_column + lane_from_tile
  B  the bomb queue is empty                                                   -> skips the bomb pass, loops A / B, TickBombs

Placements: natural; a random permutation (control: must equal natural); sorted by class inside groups of 64 / 128 / 256 / 1024
envs (what a workgroup of 4 / 8 / 16 wavefronts — or an XCD-local list — could do without global indirection; the sorted
order is rotated by the group's index so that the heavy wavefronts do not all land on the same SIMD slot); sorted globally
(per-class lists over the whole batch) with the sorted tiles dealt round-robin so heavy tiles spread over the chip; and
"dealt": the opposite of binning — every wavefront gets the same share of every class.

One launch per step (--streams 1), timed by the HIP events the library attaches to the dispatch (pom_batch_profile).  What is
NOT in these numbers: the cost of classifying and gathering (an upper bound on the gain).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import pomcpp_amd as pa  # noqa: E402
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--kind", default="ffa")
ap.add_argument("--dist", type=int, default=1)
ap.add_argument("--samples", type=int, default=24)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--gap", type=int, default=9, help="ticks played between samples")
ap.add_argument("--keys", default="X,XF,XFR,XFRB")
args = ap.parse_args()

n = args.envs
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, stream=stream.cuda_stream, streams=1)
env.make_game(pa.make_boards(n, seed=1000003, kind=args.kind))
env.step_random(1, args.dist, ticks=300)
env.sync()
base, n_pad, rec = env.device_view()
assert n_pad == n and rec == 112


class _Raw:  # the device buffer: tiles of 16 envs, [tile][dword][env in tile] (pom_packed.h)
    __cuda_array_interface__ = {"shape": (n_pad // 16, rec, 16), "typestr": "<i4", "data": (base, False), "version": 2}


tiles = torch.as_tensor(_Raw(), device=dev)


class _State:
    """the buffer as [dword][env]: clone() gathers it, copy_() scatters it back"""
    def clone(self):
        return tiles.permute(1, 0, 2).reshape(rec, n_pad).clone()

    def copy_(self, src):
        tiles.copy_(src.reshape(rec, n_pad // 16, 16).permute(1, 0, 2))


state = _State()


def classes(st):
    """next-tick class bits per env from the packed record (pom_packed.h)"""
    meta = st[62].to(torch.int64) & 0xFFFFFFFF
    meta2 = st[63].to(torch.int64) & 0xFFFFFFFF
    b_idx, b_cnt, f_idx, f_cnt = (meta >> 8) & 0xFF, (meta >> 16) & 0xFF, (meta >> 24) & 0xFF, meta2 & 0xFF
    cols = torch.arange(n_pad, device=dev)
    top_b = st[(72 + b_idx.clamp(max=19)), cols].to(torch.int64)
    top_f = st[(92 + f_idx.clamp(max=19)), cols].to(torch.int64)
    done = ((meta2 >> 8) & 1) == 1
    x = (b_cnt > 0) & (((top_b >> 16) & 0xF) == 1) & ~done
    f = (f_cnt > 0) & (((top_f >> 16) & 0xFF) == 1) & ~done
    # a restarting env plays the first tick of its game: nothing explodes or pops there
    return {"X": x, "F": f, "R": done, "B": (b_cnt == 0) & ~done}


def key_of(cl, names):
    k = torch.zeros(n_pad, dtype=torch.int64, device=dev)
    for c in names:
        k = k * 2 + cl[c].to(torch.int64)
    return k


def perm_grouped(key, group):
    """sort by key inside groups of `group` envs; the sorted order of group g is rotated by 16 * g tiles-wise"""
    g = torch.arange(n_pad, device=dev) // group
    order = torch.argsort(g * 1024 + key, stable=True)  # grouped, sorted inside
    if group > 16:
        tiles = group // 16
        pos = torch.arange(n_pad, device=dev)
        gi, within = pos // group, pos % group
        rot = (within // 16 + gi) % tiles * 16 + within % 16  # tile t of the sorted group goes to slot (t + g) % tiles
        out = torch.empty_like(order)
        out[gi * group + rot] = order
        order = out
    return order


def perm_global(key):
    order = torch.argsort(key, stable=True)
    tiles = n_pad // 16
    t = torch.arange(tiles, device=dev)
    # deal the sorted tiles round-robin over 256 "hands" so that the heavy end of the order is spread over the whole grid
    hands = 256 if tiles % 256 == 0 else 1
    dst_tile = (t % hands) * (tiles // hands) + t // hands
    out = torch.empty_like(order)
    pos = torch.arange(n_pad, device=dev)
    out[dst_tile[pos // 16] * 16 + pos % 16] = order
    return out


def perm_dealt(key):
    """the opposite of binning: the sorted order dealt out one env at a time, so that every wavefront gets the same share of
    every class (no wavefront with seven blasts while its neighbour has none)"""
    order = torch.argsort(key, stable=True)
    tiles = n_pad // 16
    i = torch.arange(n_pad, device=dev)
    out = torch.empty_like(order)
    out[(i % tiles) * 16 + i // tiles] = order
    return out


def time_tick(saved, perm, tick):
    ms = []
    for _ in range(args.reps):
        # every placement takes the same route into the state buffer (a gather into a temporary, then a copy), so that the
        # caches are in the same condition when the tick starts: a straight copy of `saved` leaves more of the state in the
        # 256 MB memory-side cache than the gather does (262,144 envs: "random" 31 % slower than natural that way)
        state.copy_(saved.index_select(1, ident if perm is None else perm))
        env.set_tick(tick)
        env.profile(True)
        env.step_random(1, args.dist, ticks=1)
        m, k = env.profile_read()
        env.profile(False)
        assert k == 1
        ms.append(m * 1e3)
    return min(ms)


ident = torch.arange(n_pad, device=dev)
keysets = args.keys.split(",")
plac = ["natural", "random"]
for ks in keysets:
    for grp in (64, 128, 256, 1024):
        plac.append(f"{ks}/group{grp}")
    plac.append(f"{ks}/global")
    plac.append(f"{ks}/dealt")
acc = {p: [] for p in plac}
frac = {c: [] for c in "XFRB"}
wave_frac = {c: [] for c in "XFR"}
tick = 300
for s in range(args.samples):
    env.set_tick(tick)
    env.step_random(1, args.dist, ticks=args.gap)
    env.sync()
    tick += args.gap
    saved = state.clone()
    cl = classes(saved)
    for c in "XFRB":
        frac[c].append(float(cl[c].float().mean()))
    for c in "XFR":
        wave_frac[c].append(float(cl[c].view(-1, 16).any(dim=1).float().mean()))
    acc["natural"].append(time_tick(saved, None, tick))
    acc["random"].append(time_tick(saved, torch.randperm(n_pad, device=dev), tick))
    for ks in keysets:
        key = key_of(cl, ks)
        for grp in (64, 128, 256, 1024):
            acc[f"{ks}/group{grp}"].append(time_tick(saved, perm_grouped(key, grp), tick))
        acc[f"{ks}/global"].append(time_tick(saved, perm_global(key), tick))
        acc[f"{ks}/dealt"].append(time_tick(saved, perm_dealt(key), tick))
    state.copy_(saved)
    env.set_tick(tick)

print(f"# {n} envs, {args.kind} boards, move distribution {args.dist}; {args.samples} sample ticks x best of {args.reps}; one launch per step")
print("# env-ticks in class:      " + "  ".join(f"{c} {np.mean(frac[c]):.3f}" for c in "XFRB"))
print("# 16-env wavefronts with >= 1 such env (natural order): " + "  ".join(f"{c} {np.mean(wave_frac[c]):.3f}" for c in "XFR"))
nat = np.mean(acc["natural"])
for p in plac:
    v = np.array(acc[p])
    print(f"{p:18s} {v.mean():7.2f} us per launch (sd {v.std():.2f})   {100 * (v.mean() / nat - 1):+6.1f} % vs natural")
