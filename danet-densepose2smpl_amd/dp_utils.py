"""DensePose-COCO point-supervision targets of one sample: the `dp_dict` that
/root/reference/utils/dp_utils.py:12-140 builds inside the dataset and IUV_Estimator.dp_uvia_losses consumes.

Host-side numpy, like the reference (it runs once per sample in the loader).  Differences in what it takes:
  * the reference decodes the 14 part masks with pycocotools (`segm_utils.GetDensePoseMask`) -- not available here; this port
    takes the decoded 256x256 label image as ann['dp_Ilabel'];
  * the left/right flip needs the licensed UV_symmetry_transforms.mat tables (`DensePoseMethods.get_symmetric_densepose`):
    pass them as the callable `symmetric(I, U, V, x, y, Ilabel)`; a flipped sample without it raises;
  * cv2.remap(..., INTER_NEAREST, BORDER_CONSTANT 0) is restated as round-half-to-even + bounds test.
PARITY UNPINNED: neither cv2 nor pycocotools exists in this image, so the reference function cannot be run here; the crop
geometry goes through augment.transform (pinned by golden g14) and the tests check the construction on analytic cases."""
import numpy as np
import torch

from . import augment

NUM_POINTS = 196          # maximum number of annotated points per person
K = 24                    # body parts


def empty_dp_dict(heatmap_size=56):
    """The all-zero dp_dict of a sample without DensePose annotation (base_dataset.py:228-236)."""
    M2 = heatmap_size ** 2
    z = np.zeros
    return {'body_uv_ann_labels': z(M2, np.int32), 'body_uv_ann_weights': z(M2, np.float32),
            'body_uv_X_points': z(NUM_POINTS, np.float32), 'body_uv_Y_points': z(NUM_POINTS, np.float32),
            'body_uv_Ind_points': z(NUM_POINTS, np.float32), 'body_uv_I_points': z(NUM_POINTS, np.float32),
            'body_uv_U_points': z(NUM_POINTS * (K + 1), np.float32), 'body_uv_V_points': z(NUM_POINTS * (K + 1), np.float32),
            'body_uv_point_weights': z(NUM_POINTS * (K + 1), np.float32)}


def _transform1(pt, center, scale, res, invert):
    out = augment.transform(torch.tensor([[pt]], dtype=torch.float64), torch.tensor([center], dtype=torch.float64),
                            torch.tensor([scale], dtype=torch.float64), res, invert=invert)
    return out[0, 0].numpy().astype(int)


def remap_nearest(img, xs, ys):
    """img[round(ys), round(xs)] with zeros outside (cv2.remap INTER_NEAREST / BORDER_CONSTANT 0)."""
    xi, yi = np.rint(xs).astype(np.int64), np.rint(ys).astype(np.int64)
    ok = (xi >= 0) & (xi < img.shape[1]) & (yi >= 0) & (yi < img.shape[0])
    out = np.zeros(xs.shape, img.dtype)
    out[ok] = img[yi[ok], xi[ok]]
    return out


def dp_annot_process(ann, heatmap_size, crop_res, center, scale, is_flipped, symmetric=None):
    """dp_utils.py:12-140.  ann: {'bbox' [x,y,w,h], 'dp_I','dp_U','dp_V','dp_x','dp_y' (point annotations, x/y in the
    256-unit box frame), 'dp_Ilabel' (decoded 256x256 part-label image)}."""
    bb = np.asarray(ann['bbox'], dtype=np.float64)
    x1s, y1s, x2s, y2s = bb[0], bb[1], bb[0] + bb[2], bb[1] + bb[3]
    ul = _transform1([1, 1], center, scale, [crop_res] * 2, 1) - 1
    br = _transform1([crop_res + 1] * 2, center, scale, [crop_res] * 2, 1) - 1
    x1, y1, x2, y2 = float(ul[0]), float(ul[1]), float(br[0]), float(br[1])
    M = int(heatmap_size)
    Ilabel = np.asarray(ann['dp_Ilabel'])
    GT_I, GT_U, GT_V = (np.array(ann[k], dtype=np.float64) for k in ('dp_I', 'dp_U', 'dp_V'))
    GT_x, GT_y = np.array(ann['dp_x'], dtype=np.float64), np.array(ann['dp_y'], dtype=np.float64)
    GT_w = np.ones(GT_I.shape, np.float32)
    if is_flipped:
        if symmetric is None:
            raise ValueError('dp_annot_process: a flipped sample needs the UV symmetry tables (symmetric=...)')
        GT_I, GT_U, GT_V, GT_x, GT_y, Ilabel = symmetric(GT_I, GT_U, GT_V, GT_x, GT_y, Ilabel)
    # the crop window sampled at M x M, in the 256-unit frame of the annotated box
    xt = ((np.arange(x1, x2, (x2 - x1) / float(M)) - x1s) * (255. / (x2s - x1s)))[:M]
    yt = ((np.arange(y1, y2, (y2 - y1) / float(M)) - y1s) * (255. / (y2s - y1s)))[:M]
    X, Y = np.meshgrid(xt, yt)
    labels = remap_nearest(Ilabel, X.astype(np.float32), Y.astype(np.float32))
    # annotated points into heat-map pixels of the crop
    GT_y = ((GT_y / 255. * (y2s - y1s)) + y1s - y1) * (float(M) / (y2 - y1))
    GT_x = ((GT_x / 255. * (x2s - x1s)) + x1s - x1) * (float(M) / (x2 - x1))
    GT_I = GT_I.copy()
    GT_I[(GT_y < 0) | (GT_y > M - 1) | (GT_x < 0) | (GT_x > M - 1)] = 0
    inside = GT_I > 0
    GT_I, GT_U, GT_V, GT_x, GT_y, GT_w = GT_I[inside], GT_U[inside], GT_V[inside], GT_x[inside], GT_y[inside], GT_w[inside]
    n = len(GT_I)
    if n > NUM_POINTS:
        raise ValueError('dp_annot_process: %d points > %d' % (n, NUM_POINTS))
    d = empty_dp_dict(M)
    d['body_uv_X_points'][:n], d['body_uv_Y_points'][:n] = GT_x, GT_y
    d['body_uv_I_points'][:n] = GT_I
    U, V = np.zeros(NUM_POINTS, np.float32), np.zeros(NUM_POINTS, np.float32)
    U[:n], V[:n] = GT_U, GT_V
    d['body_uv_U_points'], d['body_uv_V_points'] = np.tile(U, K + 1), np.tile(V, K + 1)
    wpts = np.zeros(NUM_POINTS * (K + 1), np.float32)
    for j in range(1, K + 1):                          # one block of 196 per part: 1 where the point belongs to part j
        wpts[j * NUM_POINTS:(j + 1) * NUM_POINTS] = (d['body_uv_I_points'] == j).astype(np.float32)
    d['body_uv_point_weights'] = wpts
    d['body_uv_ann_labels'] = labels.reshape(M * M).astype(np.int32)
    d['body_uv_ann_weights'] = np.ones(M * M, np.float32)
    return d
