"""tools/asm_store_waits.py on a hand-written listing: a wait that completes a store together with a load younger than it is reported (the load's
data cannot be used before the store is acknowledged: vmcnt is one in-order queue), a wait that completes a store alone is not, and a load
waited for on its own counts as a serial round trip."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("asm_store_waits", os.path.join(ROOT, "tools", "asm_store_waits.py"))
scan_mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(scan_mod)

LISTING = """
_Z6kernelv:
.LBB0_1:
	global_load_dword v1, v[2:3], off
	s_waitcnt vmcnt(0)
	global_store_dword v[4:5], v1, off
	global_load_dword v6, v[2:3], off offset:4
	s_waitcnt vmcnt(0)
	global_store_dword v[4:5], v6, off offset:4
.LBB0_2:
	global_load_dword v7, v[2:3], off offset:8
	global_load_dword v8, v[2:3], off offset:12
	global_store_dword v[4:5], v7, off offset:8
	s_waitcnt vmcnt(1)
	s_waitcnt vmcnt(0)
	s_endpgm
"""


def test_waits_that_include_store_acknowledgements_and_serial_loads():
    out, serial = scan_mod.scan(LISTING.splitlines(), "")
    hits = out["_Z6kernelv"]
    # line numbers are 1-based positions in the listing
    lines = [h[0] for h in hits]
    # the second vmcnt(0) of .LBB0_1 completes the store of line 6 and a load younger than it
    assert 8 in lines
    # vmcnt(1) in .LBB0_2 completes the two loads — and with them the store of line 9, which is older: reported;
    # the store of line 13 stays outstanding, and the last wait completes it alone (no load behind it): not reported
    assert 14 in lines and 15 not in lines
    # the first wait of the kernel completes a load alone: not a store wait, but a serial round trip
    assert 5 not in lines
    assert serial["_Z6kernelv"] == 1
