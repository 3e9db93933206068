"""image_denoising_filter_amd -- MI355X (gfx950) drop-in for the denoise hot path of
Reefufui/image_denoising_filter: bilateral (texture / linear / layer-guided), non-local means
(single frame and temporal), normalize and the u8 pack/unpack, as hand-written HIP behind the
C-ABI of include/mi_denoise.h.  Importing this package loads libmi_denoise.so; there is no
fallback path."""
from ._lib import EXPORTED, LIB_PATH, BilateralParams, NlmParams, NormalizeParams, lib  # noqa: F401
from .api import (FMT_RGBA8, FMT_RGBA32F, LAYOUT_LINEAR, LAYOUT_TEXTURE, NLM_BENCH,  # noqa: F401
                  NLM_REFERENCE, Comm, Context, DeviceBuffer, MidError, PinnedFrames, Recording, comm_create_all, comm_unique_id, load_image, save_image,
                  shard_block, shard_halo_plan, shard_launch_plan)

__all__ = ["Context", "Comm", "comm_unique_id", "comm_create_all", "shard_block", "shard_halo_plan", "shard_launch_plan", "DeviceBuffer", "PinnedFrames", "Recording", "MidError", "load_image", "save_image", "lib", "LIB_PATH", "EXPORTED",
           "BilateralParams", "NlmParams", "NormalizeParams",
           "FMT_RGBA32F", "FMT_RGBA8", "LAYOUT_TEXTURE", "LAYOUT_LINEAR", "NLM_REFERENCE", "NLM_BENCH"]
