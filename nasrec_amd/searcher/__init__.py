"""Architecture search on the engine (reference: nasrec/searcher/): candidate sub-networks of a trained supernet are scored by
fine-tuning the last layer only (eval_subnet_from_supernet.py), one candidate per GPU process."""
