"""A model of one observation-sweep launch from measured UNIT costs: does what we know about the chip explain the 1.0 ms of
k_eliminate_grouped and the 0.45 ms of k_backsub on the 1024 x 2000-line batch?  No GPU needed.

Inputs, all measured elsewhere (nothing here is fitted to the launch times):
  * work per 64-observation tile: VALU wave-instructions and fp64 matrix products per dispatch (rocprofv3 --pmc SQ_INSTS_VALU,
    SQ_INSTS_VALU_MFMA_MOPS / 64-cycle products: profiles/round5_v7_pmc_valu.txt, round5_v7_pmc_mfma.txt, round5_launch_anatomy.txt) / tiles;
  * unit costs (tools/micro/mfma_f64_bench.hip, profiles/round2_v3_micro_fp64.txt): a SIMD with TWO resident waves of fp64 vector work issues
    one v_*_f64 per 5.8 cycles (48.5 TF of the 78.6 TF data-sheet rate; one wave alone: 8.5 cycles, i.e. 0.68 of the pair's rate); a
    v_mfma_f64_16x16x4_f64 holds the SIMD's fp64 hardware for 64 cycles whatever the other wave does;
  * per chunk: ~20 us of slot time outside the tile loop (set-up, slab flush) and 13 us from the end of a workgroup to the first instruction of
    the next one on that slot (profiles/round5_launch_anatomy.txt, items 2 and 4) - of which a SIMD whose other slot is busy loses only the part
    the lone wave cannot fill;
  * the schedule: every window is cut into 6 graded chunks (70 70 20 20 10 10 of its 200 tiles), dispatched class by class onto 1024 SIMDs x 2
    wave slots, a free slot takes the next chunk of the array (lba_api.hip::plan_layout).
  * the issue arbiter serves the OLDER of a SIMD's two waves first: of two equal 70-tile chunks that start together the one in the lower slot takes
    521 us, the other 699 (same file, item 3): shares 0.573 / 0.427 of the pair's issue.
The simulation is event-driven per SIMD: two resident waves share the SIMD's issue 0.573 : 0.427 (older : younger), a lone wave advances at 0.68 of
the pair's rate.  Output: the predicted launch time against the measured one."""
import heapq

CLOCK_GHZ = 2.40                  # SQ_BUSY_CYCLES / duration of the launch (round5_launch_anatomy.txt, item 1)
PAIR_CYCLES_PER_VALU = 5.8        # two waves per SIMD, fp64 vector stream (48.5 TF chip-wide)
LONE_RATE = 5.8 / 8.5             # one wave alone delivers this fraction of the pair's rate
MFMA_CYCLES = 64.0
SIMDS, SLOTS = 1024, 2
WINDOWS, TILES = 1024, 200
CUT = (70, 70, 20, 20, 10, 10)
CHUNK_OVERHEAD_US = 20.0          # slot time of a chunk outside its tile loop
DISPATCH_GAP_US = 13.0            # end of a workgroup -> first instruction of the next one on the slot
OLDER_SHARE = 699.0 / (521.0 + 699.0)   # of the pair's issue, to the wave that has been resident longer


def simulate(valu_per_tile, mfma_per_tile, name, measured_ms):
    tile_us = (valu_per_tile * PAIR_CYCLES_PER_VALU + mfma_per_tile * MFMA_CYCLES) / (CLOCK_GHZ * 1e3)     # SIMD time of a tile at the pair's rate
    # dispatch order: the chunks of class 0 of every window, then class 1, then class 2 (two chunks per class and window)
    queue = []
    for cls in range(3):
        for w in range(WINDOWS):
            for c in (2 * cls, 2 * cls + 1):
                queue.append(CUT[c] * tile_us + CHUNK_OVERHEAD_US * 0.5)      # work in SIMD-microseconds at the pair's rate (the overhead is one wave's: half)
    queue.reverse()
    # per SIMD: remaining work of its (at most two) resident waves; a wave advances at 0.5 (two resident) or LONE_RATE * 0.5 ... in units where the
    # pair together delivers 1 SIMD-us per us
    rem = [[None, None] for _ in range(SIMDS)]
    born = [[0.0, 0.0] for _ in range(SIMDS)]
    free_at = []                                   # (time a slot becomes available, simd, slot)
    now = [0.0] * SIMDS
    busy_pair = busy_lone = 0.0

    def advance(s, t):
        nonlocal busy_pair, busy_lone
        dt = t - now[s]
        if dt <= 0:
            return
        live = [q for q in (0, 1) if rem[s][q] is not None]
        if len(live) == 2:
            old = 0 if (born[s][0], 0) <= (born[s][1], 1) else 1
            rem[s][old] -= OLDER_SHARE * dt
            rem[s][1 - old] -= (1.0 - OLDER_SHARE) * dt
            busy_pair += dt
        elif len(live) == 1:
            rem[s][live[0]] -= LONE_RATE * dt
            busy_lone += dt
        now[s] = t

    def next_finish(s):
        live = [q for q in (0, 1) if rem[s][q] is not None]
        if not live:
            return None
        if len(live) == 1:
            return now[s] + max(rem[s][live[0]], 0.0) / LONE_RATE, live[0]
        old = 0 if (born[s][0], 0) <= (born[s][1], 1) else 1
        t_old, t_new = max(rem[s][old], 0.0) / OLDER_SHARE, max(rem[s][1 - old], 0.0) / (1.0 - OLDER_SHARE)
        return (now[s] + t_old, old) if t_old <= t_new else (now[s] + t_new, 1 - old)

    events = []                                    # (time, kind, simd, slot, stamp): kind 0 = a slot is ready for work, 1 = a wave may have finished
    stamp = [0] * SIMDS
    for s in range(SIMDS):
        for q in range(SLOTS):
            heapq.heappush(events, (0.0, 0, s, q, 0))
    end = 0.0
    while events:
        t, kind, s, q, st = heapq.heappop(events)
        if kind == 1 and st != stamp[s]:
            continue                               # the SIMD's state changed since this prediction was made
        advance(s, t)
        if kind == 0:
            if queue:
                rem[s][q] = queue.pop()
                born[s][q] = t
        else:
            nf = next_finish(s)
            if nf is not None and nf[0] <= t + 1e-9:
                rem[s][nf[1]] = None
                end = max(end, t)
                heapq.heappush(events, (t + DISPATCH_GAP_US, 0, s, nf[1], 0))
        stamp[s] += 1
        nf = next_finish(s)
        if nf is not None:
            heapq.heappush(events, (nf[0], 1, s, nf[1], stamp[s]))
    ideal = WINDOWS * TILES * tile_us / SIMDS
    print("%-24s %5.0f VALU + %4.1f MFMA per tile -> %5.2f us of SIMD time per tile at the pair's rate; perfectly packed %.3f ms;"
          % (name, valu_per_tile, mfma_per_tile, tile_us, ideal / 1e3))
    print("%-24s simulated launch %.3f ms (both slots busy %.0f %% of the SIMD-time, one %.0f %%); measured %.3f ms -> model / measured = %.2f"
          % ("", end / 1e3, 100 * busy_pair / (SIMDS * end), 100 * busy_lone / (SIMDS * end), measured_ms, end / 1e3 / measured_ms))
    return end / 1e3


if __name__ == "__main__":
    tiles = WINDOWS * TILES
    # per dispatch: SQ_INSTS_VALU, matrix products (profiles/round5_v7_pmc_valu.txt; round5_launch_anatomy.txt item 1)
    simulate(272.7e6 / tiles, 7.90e6 / tiles, "k_eliminate_grouped", 0.999)
    simulate(165.7e6 / tiles, 0.0, "k_backsub", 0.450)
