'''
Fixture T9: outputs of the REAL reference loaders / writers / crop / datasets on tiny PNG files.

    python tests/golden/make_golden_formats.py

Writes tests/golden/T9_formats/{image,depth,response,ground_truth,lidar}_{0,1,2}.png (+ radar_{0,1,2}.npy) with the reference's own save_depth /
save_response (image: PIL), and T9_formats.npz with what the reference's load_image / load_depth / load_depth_with_validity_map /
load_response / random_crop / FusionNetTrainingDataset / FusionNetInferenceDataset return for them (seeds of the global numpy RNG
stored).  points_to_depth_map lives in a setup script that cannot be imported here (it imports nuscenes-devkit): the function is
cut out of the parsed file with `ast` and executed as is.
'''
import ast
import os
import random
import sys
import warnings

sys.dont_write_bytecode = True
import numpy as np
from PIL import Image

warnings.simplefilter('ignore')
sys.path.insert(0, '/root/reference/src')
import data_utils as ref_du      # noqa: E402
import datasets as ref_ds        # noqa: E402

GOLD = os.path.dirname(os.path.abspath(__file__))
DIR = os.path.join(GOLD, 'T9_formats')
H, W = 13, 18

CROPS = [(['none'], 1), (['center'], 2), (['left', 'top'], 3), (['right', 'bottom'], 4), (['horizontal'], 5),
         (['horizontal', 'vertical'], 6), (['horizontal', 'vertical'], 7), (['horizontal', 'vertical', 'anchored'], 8),
         (['horizontal', 'vertical', 'anchored'], 9), (['vertical', 'anchored', 'left'], 10), (['horizontal', 'bottom'], 11)]


def reference_points_to_depth_map():
    path = '/root/reference/setup/setup_dataset_nuscenes_with_denseGT.py'
    tree = ast.parse(open(path).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'points_to_depth_map'][0]
    ns = {'np': np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, 'exec'), ns)
    return ns['points_to_depth_map']


def main():
    os.makedirs(DIR, exist_ok=True)
    rs = np.random.RandomState(9)
    out = {}
    names = {k: [] for k in ('image', 'depth', 'response', 'ground_truth', 'lidar')}
    for i in range(3):
        img = rs.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
        p = os.path.join(DIR, 'image_%d.png' % i)
        Image.fromarray(img).save(p)
        names['image'].append(p)
        for kind in ('depth', 'response', 'ground_truth', 'lidar'):
            dense = {'depth': 0.25, 'response': 0.25, 'ground_truth': 0.6, 'lidar': 0.1}[kind]
            z = (rs.rand(H, W) * 120.0 * (rs.rand(H, W) < dense)).astype(np.float32)
            if kind == 'depth' and i == 0:
                z[0, :6] = [0.0, 1.5, 255.99, 256.0, 300.0, -1.0]      # 16-bit ceiling and the negative wrap of np.uint32
            p = os.path.join(DIR, '%s_%d.png' % (kind, i))
            if kind == 'response':
                z = (z / 120.0).astype(np.float32)                     # responses are sigmoid outputs
                out['saved_response_%d' % i] = z
                ref_du.save_response(z, p)
            else:
                out['saved_%s_%d' % (kind, i)] = z
                ref_du.save_depth(z, p)
            names[kind].append(p)

    for i in range(3):
        out['load_image_hwc_%d' % i] = ref_du.load_image(names['image'][i])
        out['load_image_chw_norm_%d' % i] = ref_du.load_image(names['image'][i], normalize=True, data_format='CHW')
        out['load_depth_%d' % i] = ref_du.load_depth(names['depth'][i])
        z, v = ref_du.load_depth_with_validity_map(names['depth'][i], data_format='CHW')
        out['load_depth_v_z_%d' % i], out['load_depth_v_v_%d' % i] = z, v
        out['load_response_%d' % i] = ref_du.load_response(names['response'][i], data_format='HWC')
        out['response_as_depth_%d' % i] = ref_du.load_depth(names['response'][i], data_format='CHW')
        out['pixels_depth_%d' % i] = np.array(Image.open(names['depth'][i]))
        out['pixels_response_%d' % i] = np.array(Image.open(names['response'][i]))

    # random_crop on sample 0 (image + depth, CHW)
    image = ref_du.load_image(names['image'][0], data_format='CHW')
    depth = ref_du.load_depth(names['depth'][0], data_format='CHW')
    shape = (8, 11)
    for ci, (crop_type, seed) in enumerate(CROPS):
        np.random.seed(seed)
        a, b = ref_ds.random_crop([image, depth], shape, crop_type)
        out['crop_image_%d' % ci], out['crop_depth_%d' % ci] = a, b
        out['crop_next_rand_%d' % ci] = np.random.rand(1)        # pins how much of the RNG stream was consumed

    # datasets
    np.random.seed(21)
    ds = ref_ds.FusionNetTrainingDataset(names['image'], names['depth'], names['response'], names['ground_truth'], names['lidar'],
                                         shape=shape, random_crop_type=['horizontal', 'vertical', 'anchored'])
    for rep in range(2):
        for i in range(len(ds)):
            for j, t in enumerate(ds[i]):
                out['train_ds_%d_%d_%d' % (rep, i, j)] = t
    ds = ref_ds.FusionNetTrainingDataset(names['image'], names['depth'], names['response'], names['ground_truth'], names['lidar'])
    for j, t in enumerate(ds[1]):
        out['train_ds_full_1_%d' % j] = t
    ds = ref_ds.FusionNetInferenceDataset(names['image'], names['depth'], names['response'], names['ground_truth'])
    for j, t in enumerate(ds[2]):
        out['infer_ds_2_%d' % j] = t
    ds = ref_ds.FusionNetInferenceDataset(names['image'], names['depth'], names['response'], [None] * 3)
    out['infer_ds_nogt_len'] = np.array([len(ds[0])])

    # RadarNet datasets: radar_{0,1,2}.npy = many points / fewer than sampled / a single 1-D point
    radar = [np.stack([rs.rand(9) * (W - 1), rs.rand(9) * (H - 1), rs.rand(9) * 60 + 2], 1),
             np.stack([rs.rand(2) * (W - 1), rs.rand(2) * (H - 1), rs.rand(2) * 60 + 2], 1),
             np.array([7.3, 5.1, 33.0])]
    names['radar'] = []
    for i, r in enumerate(radar):
        p = os.path.join(DIR, 'radar_%d.npy' % i)
        np.save(p, r)
        names['radar'].append(p)
    for tag, prob in (('radar', 0.0), ('lidar', 1.0), ('mixed', 0.5)):
        np.random.seed(31)
        random.seed(32)
        ds = ref_ds.RadarNetTrainingDataset(names['image'], names['radar'], names['ground_truth'], patch_size=(9, 6),
                                            total_points_sampled=4, sample_probability_of_lidar=prob)
        for rep in range(2):
            for i in range(len(ds)):
                for j, t in enumerate(ds[i]):
                    out['radarnet_train_%s_%d_%d_%d' % (tag, rep, i, j)] = t
        out['radarnet_train_%s_next' % tag] = np.array([np.random.rand(), random.random()])
    ds = ref_ds.RadarNetInferenceDataset(names['image'], names['radar'], names['ground_truth'])
    for i in range(3):
        for j, t in enumerate(ds[i]):
            out['radarnet_infer_%d_%d' % (i, j)] = t
    out['radarnet_infer_nogt_len'] = np.array([len(ref_ds.RadarNetInferenceDataset(names['image'], names['radar'])[0])])

    # points_to_depth_map
    p2d = reference_points_to_depth_map()
    n = 300
    pts = np.stack([rs.rand(n) * (W - 1), rs.rand(n) * (H - 1)]).astype(np.float32)
    pts[:, :8] = np.array([[2.5, 3.5, 0.5, 1.5, 4.49, 4.51, -0.4, 16.6], [2.5, 3.5, 6.5, 7.5, 0.0, 0.0, 0.4, 11.7]], np.float32)
    pts[:, 8] = [-1.0, -2.0]                                       # numpy wraps negative indices: column W - 1, row H - 2
    pts[:, 100:140] = pts[:, 20:60]                                # later duplicates overwrite earlier points
    dep = (rs.rand(n) * 80.0 + 1.0).astype(np.float32)
    out['p2d_points'], out['p2d_depth'] = pts, dep
    out['p2d_map'] = p2d(pts, dep, np.zeros((H, W, 3), np.uint8))
    pts64 = pts.astype(np.float64) + 1e-9                          # float64 points: ties decided in double precision
    out['p2d_points64'] = pts64
    out['p2d_map64'] = p2d(pts64, dep, np.zeros((H, W, 3), np.uint8))

    out['crop_shape'] = np.array(shape)
    np.savez_compressed(os.path.join(GOLD, 'T9_formats.npz'), **out)
    print('wrote', len(out), 'arrays and', sum(len(v) for v in names.values()), 'PNG files')


if __name__ == '__main__':
    main()
