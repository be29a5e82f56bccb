'''
Stage 1 -> stage 2 glue of the method: from an image and its radar points to the dense radar depth / response maps that
FusionNet consumes -- radarnet_main.forward (src/radarnet_main.py:534-591) on the HIP path: RadarNetModel.forward on the
edge-padded image for all points at once, then rcf_radar_scatter_logits (sigmoid and the 0.5 threshold, taken on the logit's sign, max /
argmax over the points' canvases,
and the reference's int64 index -> depth replacement chain, quirks included).
'''
import torch

from . import ops


def radarnet_forward(model, image, radar_points, bounding_boxes_list):
    '''
    Arg(s):
        model : RadarNetModel
        image : torch.Tensor[float32]
            1 x 3 x H x W image (not padded)
        radar_points : torch.Tensor[float32]
            K x 3 (or 1 x K x 3) points (x, y, z) with x already shifted by patch_width // 2, as radarnet_main.run prepares them
            (src/radarnet_main.py:638-642)
        bounding_boxes_list : list[torch.Tensor[float32]]
            [K x 4] boxes in the padded image's coordinates (src/radarnet_main.py:643-650)
    Returns:
        torch.Tensor[float32] : 1 x H x W output depth (radar depth assigned to the pixels of the winning point)
        torch.Tensor[float32] : 1 x H x W maximum response
    '''
    patch = model.input_patch_size_image
    pad = int(patch[1]) // 2
    # torchvision.transforms.functional.pad(image, (pad, 0, pad, 0), padding_mode='edge') (:540-543)
    image_p = torch.nn.functional.pad(image, (pad, pad, 0, 0), mode='replicate')
    if radar_points.dim() == 3:
        radar_points = torch.squeeze(radar_points, dim=0)
    height, width = image.shape[-2], image.shape[-1]
    crop_height = height - int(patch[0])
    with torch.no_grad():
        # logits, not torch.sigmoid(logits): the 0.5 threshold of :563-567 is taken on the sign of the logit inside the scatter kernel
        logits = model.forward(image=image_p, point=radar_points, bounding_boxes=bounding_boxes_list, return_logits=True)
    depth, resp = ops.radar_scatter(logits[:, 0].contiguous(), radar_points.contiguous().to(torch.float32), width,
                                    strict_reference=True, logits=True)
    if crop_height > 0:   # crops cover the bottom patch_height rows (:566-569); nothing is predicted above them
        top = torch.zeros((crop_height, width), dtype=torch.float32, device=depth.device)
        depth, resp = torch.cat([top, depth], 0), torch.cat([top, resp], 0)
    return depth.unsqueeze(0), resp.unsqueeze(0)
