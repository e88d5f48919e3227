#!/usr/bin/env python3
"""Static scan of a gfx950 assembly listing (hipcc -S --cuda-device-only) for waits that include STORE acknowledgements.

vmcnt counts loads and stores in ONE in-order queue: `s_waitcnt vmcnt(N)` behind a store waits for that store to be acknowledged by
memory unless at least N younger operations follow it.  A load issued after stores (element-by-element epilogues: load, use, store, load
...) therefore costs a store round trip per element.  The scan is linear (control flow ignored, every instruction in listing order), so
it over- and under-counts around branches; it is a pointer to where to read the listing, not a proof.

    python tools/asm_store_waits.py file.s [kernel-substring]
"""
import re
import sys


def scan(lines, want):
    kern, out = None, {}
    serial = {}
    global serial_at
    serial_at = {}
    q = []  # outstanding vector-memory operations, oldest first: (kind, line)
    label = ""
    for ln, raw in enumerate(lines, 1):
        s = raw.strip()
        if not s or s.startswith(";"):
            continue
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kern, q, label = m.group(1), [], ""
            continue
        if kern is None or (want and want not in kern):
            continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            label = m.group(1)
            continue
        op = s.split()[0]
        if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
            q.append(("L", ln))
        elif op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic")):
            q.append(("S", ln))
        elif op == "s_waitcnt" and "vmcnt" in s:
            n = int(re.search(r"vmcnt\((\d+)\)", s).group(1))
            done, q = (q[:-n], q[-n:]) if n else (q, [])
            if len(done) > len(q) + n:
                pass
            st = [x for x in done if x[0] == "S"]
            ld = [x for x in done if x[0] == "L"]
            if st and ld and max(x[1] for x in ld) > min(x[1] for x in st):  # a load younger than a store is waited for: the store too
                out.setdefault(kern, []).append((ln, label, n, len(st), len(ld)))
            if len(done) == 1 and not q and done[0][0] == "L":  # a wait for ONE load with nothing else in flight: a serial round trip
                serial[kern] = serial.get(kern, 0) + 1
                serial_at.setdefault(kern, []).append((ln, label, done[0][1]))
        elif op == "s_endpgm":
            kern = None
    return out, serial


def main():
    lines = open(sys.argv[1]).read().splitlines()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    out, serial = scan(lines, want)
    for kern in sorted(set(out) | set(serial)):
        hits = out.get(kern, [])
        print("%s: %d waits that include store acknowledgements, %d waits for a single load with nothing else in flight" % (kern, len(hits), serial.get(kern, 0)))
        if len(sys.argv) > 4:  # a fourth argument: list the serial round trips too (line of the wait, label, line of the load)
            for ln, label, lload in serial_at.get(kern, []):
                print("   serial: wait at line %6d after %-12s for the load at line %d" % (ln, label, lload))
        for ln, label, n, ns, nl in hits[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
            print("   line %6d  after %-12s vmcnt(%d): %d stores, %d loads completed here" % (ln, label, n, ns, nl))


if __name__ == "__main__":
    main()
