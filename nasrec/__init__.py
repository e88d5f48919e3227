"""Import-path shim: lets the reference's own scripts (`from nasrec.supernet.supernet import SuperNet`, …) resolve to
the MI355X engine's implementation of the same API (nasrec_amd).  Only the hot-path modules are aliased
(INTEGRATION.md §1); everything else of the reference's `nasrec` package (CLIs, data pipes, searcher) is out of scope."""
