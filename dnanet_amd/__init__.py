"""dnanet_amd -- MI355X-native (gfx950) implementation of DNANet's U-Net peak-segmentation hot path."""
__version__ = "0.1.0"
