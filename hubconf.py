"""torch.hub entry point with the reference's name and contract (hubconf.py:1-11 there):

    model = torch.hub.load(<this repo>, 'vgg16_netvlad', pretrained=False, source='local')
    desc  = model.cuda().eval()(images)          # [N][3][H][W] float32 -> [N][4096], unit rows

The returned EmbedNetPCA owns the same parameters under the same state-dict keys as the reference's,
and its forward runs on the MI355X HIP kernels (openibl_amd)."""
dependencies = ['torch']
import torch
from ibl import models

_RELEASE_URL = 'https://github.com/yxgeee/OpenIBL/releases/download/v0.1.0-beta/vgg16_netvlad.pth'


def vgg16_netvlad(pretrained=False):
    base_model = models.create('vgg16', pretrained=False)
    pool_layer = models.create('netvlad', dim=base_model.feature_dim)
    model = models.create('embednetpca', base_model, pool_layer)
    if pretrained:
        model.load_state_dict(torch.hub.load_state_dict_from_url(
            _RELEASE_URL, map_location=torch.device('cpu')))
    return model
