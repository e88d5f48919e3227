#!/usr/bin/env python3
"""Offline model of the persistent step's dispatch (round 6): workgroups enter in index order while one of the chip's slots is free (768 with
the Transformer backward's 52 KB of LDS, 1024 without), a workgroup that has entered runs once its item's dependencies are done (+ the measured
2.5 us seam) for the item's measured per-workgroup duration, and holds its slot from entry to end.  Fed with the per-item stamps of
tools/persist_timeline.py (NASREC_TIMELINE_JSON), it reproduces the measured launch durations of two of the three launches of the cfg-2 step to
within 1 us (the third to 13 %), and answers what a better ITEM ORDER could buy: greedy list schedules and 6000 adjacent-swap steps find nothing
for the first two launches (both already sit on their critical paths: 36.4 / 85.0 us of dependent items against 38.9 / 91.2 measured) and 13 %
for the third.  The critical path itself is the limit: its Transformer bodies run 25 - 28 us beside the workgroups the launch packs around them,
against 17 - 19 us as items of a level launch that the balancing pass keeps light (profiles/r06_persist_timeline_nothrottle.txt).

    NASREC_PERSIST_THROTTLE=0 NASREC_TIMELINE_JSON=/tmp/tl.json python tools/persist_timeline.py     # on the GPU box
    python tools/persist_sim.py /tmp/tl.json                                                          # anywhere"""
import heapq
import json
import random
import sys


def simulate(order_items, slots, seam=1.5, arrive=1.0):
    """order_items: list of dicts(nwg, dur, deps(list of indices into this list)) -> (makespan, per item done/ready/enter)"""
    free = [0.0]*slots
    heapq.heapify(free)
    done=[]; info=[]
    last_entry=0.0
    for it in order_items:
        ready = max([done[j] for j in it['deps']] + [0.0]) + (seam if it['deps'] else 0.5)
        fin_max=0.0; e0=None
        for w in range(it['nwg']):
            t=heapq.heappop(free)
            entry=max(t,last_entry); last_entry=entry
            if e0 is None: e0=entry
            start=max(entry+0.7, ready)
            fin=start+it['dur']
            fin_max=max(fin_max,fin)
            heapq.heappush(free,fin)
        done.append(fin_max+arrive)
        info.append((e0,last_entry,ready,fin_max))
    return max(done), info


def main(prefix):
    random.seed(1)
    si = 0
    while True:
        try:
            seg = json.load(open("%s.%d" % (prefix, si)))
        except OSError:
            break
        n = len(seg)
        for it in seg:
            it["dur"] = max(it["body_end"] - max(it["ready"], it["enter_last"]), 0.8)
        slots = 768 if any(it["kind"] == "MHA_BWD" for it in seg) else 1024

        def run(order):
            pos = {k: i for i, k in enumerate(order)}
            return simulate([dict(nwg=seg[k]["nwg"], dur=seg[k]["dur"], deps=[pos[j] for j in seg[k]["deps"]]) for k in order], slots, seam=2.5)[0]
        cur = list(range(n))
        cv = run(cur)
        fin = [0.0] * n
        for k in range(n):
            fin[k] = max([fin[j] for j in seg[k]["deps"]] + [0.0]) + seg[k]["dur"] + 2.5
        print("launch %d: measured %.1f us, model %.1f us in the measured order, critical path %.1f us" % (si, max(it["done"] for it in seg), cv, max(fin)))
        for _ in range(6000):
            i = random.randrange(n - 1)
            a, b = cur[i], cur[i + 1]
            if a in seg[b]["deps"]:
                continue
            cur[i], cur[i + 1] = b, a
            v = run(cur)
            if v <= cv:
                cv = v
            else:
                cur[i], cur[i + 1] = a, b
        print("   best order found by adjacent swaps: %.1f us" % cv)
        si += 1


if __name__ == "__main__":
    main(sys.argv[1])
