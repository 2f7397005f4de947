from openibl_amd.models import NetVLAD, EmbedNet, EmbedNetPCA, EmbedRegionNet

__all__ = ['NetVLAD', 'EmbedNet', 'EmbedNetPCA', 'EmbedRegionNet']
