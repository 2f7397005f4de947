from openibl_amd.models import VGG, vgg16

__all__ = ['VGG', 'vgg16']
