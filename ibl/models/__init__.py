"""ibl.models — same factory surface as the reference (ibl/models/__init__.py:7-53)."""
from openibl_amd.models import (VGG, vgg16, NetVLAD, EmbedNet, EmbedNetPCA, EmbedRegionNet,
                                names, create)

__all__ = ['VGG', 'vgg16', 'NetVLAD', 'EmbedNet', 'EmbedNetPCA', 'EmbedRegionNet', 'names',
           'create']
