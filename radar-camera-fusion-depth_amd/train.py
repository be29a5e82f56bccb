'''
One FusionNet training step as the reference's loop body does it (src/fusionnet_main.py:369-399):
forward -> compute_loss (l1 + lidar) -> zero_grad -> backward -> optimizer.step.  Used by bench.py, smoke() and
the parity tests; the reference's own loop can equally drive FusionNetModel directly.
'''

from . import synth
from .fusionnet_model import FusionNetModel
from .optim import FusedAdam


def build_model(cfg=None, device='cuda', weight_initializer='kaiming_uniform', deconv_type='up'):
    '''FusionNetModel with the shipped flags (bash/train_fusionnet_nuscenes.sh:27-40; deconv 'up' per src/fusionnet_main.py:190).'''
    cfg = synth.PUBLISHED if cfg is None else cfg
    return FusionNetModel(
        input_channels_image=cfg['input_channels_image'],
        input_channels_depth=cfg['input_channels_depth'],
        encoder_type=['fusionnet18', 'batch_norm'],
        n_filters_encoder_image=cfg['n_filters_encoder_image'],
        n_filters_encoder_depth=cfg['n_filters_encoder_depth'],
        fusion_type='weight_and_project',
        decoder_type=['multiscale', 'batch_norm'],
        n_resolution_decoder=1,
        n_filters_decoder=cfg['n_filters_decoder'],
        deconv_type=deconv_type,
        activation_func='leaky_relu',
        weight_initializer=weight_initializer,
        min_predict_depth=1.0,
        max_predict_depth=100.0,
        device=device)


def make_optimizer(model, lr=1e-3, weight_decay=0.0):
    return FusedAdam([{'params': model.parameters(), 'weight_decay': weight_decay}], lr=lr)


def train_step(model, optimizer, image, input_depth, ground_truth, lidar_map, w_lidar_loss=2.0, outlier_removal=None):
    '''outlier_removal: a net_utils.OutlierRemoval applied to the ground truth first (src/fusionnet_main.py:377-378).'''
    output_depth = model.forward(image=image, input_depth=input_depth)
    if outlier_removal is not None:
        ground_truth = outlier_removal.remove_outliers(ground_truth)
    loss, loss_info = model.compute_loss(
        image=image, output_depth=output_depth, ground_truth=ground_truth, lidar_map=lidar_map,
        loss_func='l1', w_smoothness=0.0, loss_smoothness_kernel_size=-1,
        validity_map_loss_smoothness=None, w_lidar_loss=w_lidar_loss)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss, loss_info, output_depth
