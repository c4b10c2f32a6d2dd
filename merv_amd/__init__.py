"""merv_amd: MI355X-native multi-encoder video forward path of MERV (hand-written HIP behind a C ABI)."""
__all__ = ["_lib", "ops", "encoder", "projector"]
