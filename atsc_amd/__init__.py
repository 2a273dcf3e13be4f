"""atsc_amd -- MI355X-native ATSC compression core (per-frame auto-compressor path).

The product is libatsc_hip.so (hand-written gfx950 kernels behind the C ABI in
include/atsc_hip.h).  This package only loads it and moves pointers around."""
from . import capi  # noqa: F401
from .capi import AUTO, CONSTANT, FFT, IDW, NOOP, POLYNOMIAL, RLE, AtscError  # noqa: F401
from .engine import Context, DPlan, Plan, bro_open, bro_prefix, chunk_sizes, clean_data  # noqa: F401
from .stream import (CompressedStream, bro_read_file, compress_data, csv_read, decompress_data,  # noqa: F401
                     wbro_from_bytes, wbro_read, wbro_to_bytes, wbro_write)
from .vsri import Metric, Vsri, day_elapsed_seconds, read_samples_from_csv_file, write_samples_to_csv_file  # noqa: F401
