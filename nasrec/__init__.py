"""Import-path shim: lets the reference's own scripts (`from nasrec.supernet.supernet import SuperNet`, …) resolve to
the MI355X engine's implementation of the same API (nasrec_amd).  The hot-path modules and the training harness around them (main_train.py, train_supernet.py,
utils/{train_utils,data_pipes,io_utils,lr_schedule}.py — SURVEY §8f-1) are aliased; the searcher is out of scope."""
