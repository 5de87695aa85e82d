"""quokka_amd — MI355X-native implementation of Quokka's PPM+HLLC hydro / M1 radiation hot path.

The product is the HIP library behind include/quokka_amd.h (quokka_amd/csrc) and the C++ host
mirror of the reference's operator surface (quokka_amd/host).  The Python modules here are
plumbing for the test-suite and the bench harness (device memory, streams, torch.distributed).
"""
__all__ = ["capi", "multifab", "hydro_system"]
