"""DensePose-COCO point-supervision targets of one sample: the `dp_dict` that
/root/reference/utils/dp_utils.py:12-140 builds inside the dataset and IUV_Estimator.dp_uvia_losses consumes.

Host-side numpy, like the reference (it runs once per sample in the loader), with the reference's input contract: the
DensePose-COCO annotation as it lies in the json -- ann['bbox'], the five point lists dp_I / dp_U / dp_V / dp_x / dp_y and
ann['dp_masks'], 14 COCO run-length-encoded 256x256 part masks (or [] for an absent part).
  * the reference decodes the masks with pycocotools (`segm_utils.GetDensePoseMask`, /root/reference/utils/segms.py:34-40; a
    third-party dependency that is neither vendored nor pinned by the reference and absent from this image): `rle_decode`
    restates the published algorithm of its maskApi.c (rleFrString + rleDecode: 5-bit groups offset by 48, bit 0x20 =
    continuation, sign extension from bit 0x10, counts from the third on stored as differences to the count two back;
    column-major runs alternating 0 / 1 starting with 0).  A decoded label image may be passed instead as ann['dp_Ilabel'];
  * the left / right flip is DensePoseSymmetry.get_symmetric_densepose (/root/reference/utils/densepose_methods.py:31-59) over
    the licensed UV_symmetry_transforms.mat tables (24 U / V lookup images of 256 x 256), loaded from a path or passed as arrays;
  * cv2.remap(..., INTER_NEAREST, BORDER_CONSTANT 0) is restated as round-half-to-even + bounds test.
Parity: get_symmetric_densepose and the arithmetic of dp_annot_process are pinned by golden g15 (tests/golden/make_golden.py
runs the reference's own functions on synthetic tables and annotations); the two third-party primitives (pycocotools decode,
cv2.remap) are NOT -- neither library exists here, the generator plugs this file's restatements into the reference for them."""
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


def rle_decode(rle):
    """One COCO run-length-encoded mask -> uint8 [h, w] (pycocotools maskApi.c rleFrString / rleDecode).  rle: {'size': [h, w],
    'counts': compressed str / bytes, or the uncompressed list of run lengths}."""
    h, w = int(rle['size'][0]), int(rle['size'][1])
    c = rle['counts']
    if isinstance(c, (list, tuple, np.ndarray)):
        cnts = [int(v) for v in c]
    else:
        if isinstance(c, str):
            c = c.encode('ascii')
        cnts, p = [], 0
        while p < len(c):
            x, k, more = 0, 0, True
            while more:
                v = c[p] - 48
                x |= (v & 0x1f) << (5 * k)
                more = bool(v & 0x20)
                p += 1
                k += 1
                if not more and (v & 0x10):
                    x |= -1 << (5 * k)
            if len(cnts) > 2:
                x += cnts[-2]
            cnts.append(x)
    if sum(cnts) != h * w or min(cnts, default=0) < 0:
        raise ValueError('rle_decode: the runs cover %d of %d pixels' % (sum(cnts), h * w))
    flat = np.repeat(np.arange(len(cnts), dtype=np.uint8) & 1, cnts)
    return flat.reshape(w, h).T.copy()                     # runs go down the columns


def rle_encode(mask):
    """uint8 [h, w] -> {'size', 'counts': compressed bytes} (maskApi.c rleEncode / rleToString); the inverse of rle_decode,
    used by the tests and the golden generator to build annotations."""
    m = (np.asarray(mask) != 0).T.reshape(-1)
    edges = np.flatnonzero(np.diff(m.astype(np.int8))) + 1
    cnts = np.diff(np.concatenate([[0], edges, [m.size]])).tolist()
    if m.size and m[0]:
        cnts = [0] + cnts
    out = bytearray()
    for i, x in enumerate(cnts):
        if i > 2:
            x -= cnts[i - 2]
        more = True
        while more:
            v = x & 0x1f
            x >>= 5
            more = (x != -1) if (v & 0x10) else (x != 0)
            out.append((v | 0x20 if more else v) + 48)
    return {'size': [int(mask.shape[0]), int(mask.shape[1])], 'counts': bytes(out)}


def get_densepose_mask(polys):
    """The 14 part masks of an annotation -> one 256 x 256 label image, later parts painted over earlier ones
    (/root/reference/utils/segms.py:34-40)."""
    lab = np.zeros((256, 256))
    for i in range(1, 15):
        if polys[i - 1]:
            lab[rle_decode(polys[i - 1]) > 0] = i
    return lab


class DensePoseSymmetry(object):
    """Mirror-symmetric DensePose labels (/root/reference/utils/densepose_methods.py:15-59, the part of DensePoseMethods the
    loader uses).  tables: path of UV_symmetry_transforms.mat, or {'U_transforms', 'V_transforms'}: 24 lookup images of
    256 x 256 each (object arrays [1, 24] as scipy.io.loadmat returns them, or arrays [24, 256, 256])."""
    MASK_SYMMETRY = [0, 1, 3, 2, 5, 4, 7, 6, 9, 8, 11, 10, 13, 12, 14]                 # 14 coarse parts: left <-> right
    INDEX_SYMMETRY = [1, 2, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15, 18, 17, 20, 19, 22, 21, 24, 23]       # 24 surface patches

    def __init__(self, tables):
        if isinstance(tables, str):
            from scipy.io import loadmat
            tables = loadmat(tables)

        def as_list(t):
            t = np.asarray(t)
            return [np.asarray(t[0, i]) for i in range(24)] if t.dtype == object else [t[i] for i in range(24)]
        self.u_tab, self.v_tab = as_list(tables['U_transforms']), as_list(tables['V_transforms'])

    def get_symmetric_densepose(self, I, U, V, x, y, mask):
        I, U, V = np.asarray(I), np.asarray(U), np.asarray(V)
        labels, u_sym, v_sym = np.zeros(I.shape), np.zeros(U.shape), np.zeros(V.shape)
        for i in range(24):
            sel = np.where(I == i + 1)
            if len(sel[0]) == 0:
                continue
            labels[sel] = self.INDEX_SYMMETRY[i]
            ul, vl = (U[sel] * 255).astype(np.int64), (V[sel] * 255).astype(np.int64)
            v_sym[sel] = self.v_tab[i][vl, ul]
            u_sym[sel] = self.u_tab[i][vl, ul]
        flip = np.fliplr(np.asarray(mask))
        mask_sym = np.zeros(flip.shape)
        for i in range(14):
            mask_sym[flip == i + 1] = self.MASK_SYMMETRY[i + 1]
        return labels, u_sym, v_sym, flip.shape[1] - np.asarray(x), y, mask_sym

    __call__ = get_symmetric_densepose


def synthetic_symmetry_tables():
    """Closed-form stand-ins for the licensed UV_symmetry_transforms.mat lookup images ([24, 256, 256] each, values in [0, 1]):
    what the golden generator (g15) gives the reference and the tests give DensePoseSymmetry.  Not anatomically meaningful."""
    i, v, u = np.meshgrid(np.arange(24), np.arange(256), np.arange(256), indexing='ij')
    u_tab = (((i * 37 + v * 11 + u * 201) % 256) / 255.0).astype(np.float32)
    v_tab = (((i * 91 + v * 163 + u * 7 + 5) % 256) / 255.0).astype(np.float32)
    return u_tab, v_tab


def dp_annot_process(ann, heatmap_size, crop_res, center, scale, is_flipped, symmetric=None):
    """dp_utils.py:12-140.  ann: {'bbox' [x,y,w,h], 'dp_I','dp_U','dp_V','dp_x','dp_y' (point annotations, x/y in the
    256-unit box frame), 'dp_masks' (14 RLE part masks) or 'dp_Ilabel' (the decoded 256x256 part-label image)};
    symmetric: a DensePoseSymmetry (or any callable with get_symmetric_densepose's signature), needed for flipped samples."""
    bb = np.asarray(ann['bbox'], dtype=np.float64)
    x1s, y1s, x2s, y2s = bb[0], bb[1], bb[0] + bb[2], bb[1] + bb[3]
    ul = _transform1([1, 1], center, scale, [crop_res] * 2, 1) - 1
    br = _transform1([crop_res + 1] * 2, center, scale, [crop_res] * 2, 1) - 1
    x1, y1, x2, y2 = float(ul[0]), float(ul[1]), float(br[0]), float(br[1])
    M = int(heatmap_size)
    Ilabel = np.asarray(ann['dp_Ilabel']) if 'dp_Ilabel' in ann else get_densepose_mask(ann['dp_masks'])
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
