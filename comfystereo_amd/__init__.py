"""comfystereo_amd -- MI355X-native depth-to-stereo engine, drop-in for ComfyStereo's Stereo Image Node.

Import is cheap and GPU-free; the HIP library is loaded on first use (see _native.lib()).
"""
__version__ = "0.1.0"
