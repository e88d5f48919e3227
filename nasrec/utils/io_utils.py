from nasrec_amd.utils.io_utils import (create_dir, dump_json, dump_pickle_data, load_json, load_model_checkpoint, load_pickle_data,  # noqa: F401
                                       save_model_checkpoint)
