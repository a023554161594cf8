"""Dataset loader with the reference's on-disk layout and return dict.

Reference: /root/reference/NeRFs/DFANeRF/load_audface.py:11-176.  Files under <basedir>:
  transforms_{train,val}[_ba].json  {focal_len, cx, cy, frames:[{img_id, aud_id, transform_matrix 4x4, face_rect}]}
  <aud_file>  torch tensor [N,512];  <exp_file>  dict with 'exp_o' [N,64];  bc.jpg;  head_imgs/, com_imgs/, ori_imgs/
imageio / cv2 are not needed: JPEGs are read with PIL."""
import json
import os

import numpy as np
import torch


def _imread(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert('RGB'))


def _features(basedir, aud_file, exp_file, exp_offset=0):
    exp = torch.load(os.path.join(basedir, exp_file), weights_only=False)['exp_o'].numpy()[exp_offset:]
    aud = torch.load(os.path.join(basedir, aud_file), weights_only=False).cpu().numpy()
    return aud, exp


def load_audface_data_split(basedir, testskip=1, test_file=None, aud_file=None, exp_file='face.pt', no_com=False,
                            all_speaker=False, use_ori=False, use_ba=False, test_offset=0):
    if test_file:
        with open(os.path.join(basedir, test_file)) as fp:
            meta = json.load(fp)
        aud_f, exp_f = _features(basedir, aud_file, exp_file, test_offset)
        frames = meta['frames'][::testskip]
        pick = lambda feats, fid: feats[min(fid, feats.shape[0] - 1)]
        bc_img = _imread(os.path.join(basedir, 'bc.jpg'))
        return {'poses': np.array([f['transform_matrix'] for f in frames]).astype(np.float32),
                'auds': np.array([pick(aud_f, f['img_id']) for f in frames]).astype(np.float32),
                'bc_img': bc_img,
                'hwfcxy': [bc_img.shape[0], bc_img.shape[1], float(meta['focal_len']), float(meta['cx']),
                           float(meta['cy'])],
                'exp': np.array([pick(exp_f, f['img_id']) for f in frames]).astype(np.float32)}

    aud_f, exp_f = _features(basedir, aud_file, exp_file)
    cols = {k: [] for k in ('imgs', 'imgs_com', 'imgs_ori', 'poses', 'auds', 'exps', 'rects')}
    counts = [0]
    meta = None
    for s in ('train', 'val'):
        name = 'transforms_{}_ba.json'.format(s) if use_ba else 'transforms_{}.json'.format(s)
        with open(os.path.join(basedir, name), 'r') as fp:
            meta = json.load(fp)
        skip = 1 if (s == 'train' or testskip == 0) else testskip
        frames = meta['frames'][::skip]
        for f in frames:
            fid = '{:06d}.jpg'.format(f['img_id'])
            cols['imgs'].append(os.path.join(basedir, 'head_imgs', fid))
            cols['imgs_com'].append(os.path.join(basedir, 'com_imgs', fid))
            cols['imgs_ori'].append(os.path.join(basedir, 'ori_imgs', fid))
            cols['poses'].append(np.array(f['transform_matrix']))
            cols['auds'].append(aud_f[min(f['aud_id'], aud_f.shape[0] - 1)])
            cols['exps'].append(exp_f[min(f['img_id'], exp_f.shape[0] - 1)])
            cols['rects'].append(np.array(f['face_rect'], dtype=np.int32))
        counts.append(counts[-1] + len(frames))
    n = counts[-1]
    bc_img = _imread(os.path.join(basedir, 'bc.jpg'))
    speak = np.zeros(n, dtype=np.int32)
    if all_speaker:
        speak += 1
    else:
        st = np.load(os.path.join(basedir, 'speak_time.npy'))
        for k in range(st.shape[0]):
            speak[np.arange(int(st[k, 0] * 30) + 1, int(st[k, 1] * 30) - 1)] = 1
    return {'imgs': np.array(cols['imgs']),
            'imgs_com': None if no_com else np.array(cols['imgs_com']),
            'poses': np.array(cols['poses']).astype(np.float32),
            'auds': np.array(cols['auds']).astype(np.float32),
            'bc_img': bc_img,
            'hwfcxy': [bc_img.shape[0], bc_img.shape[1], float(meta['focal_len']), float(meta['cx']),
                       float(meta['cy'])],
            'sample_rects': np.array(cols['rects']),
            'i_split': [np.arange(counts[k], counts[k + 1]) for k in range(2)],
            'speak_frames': speak,
            'exp': np.array(cols['exps']).astype(np.float32),
            'imgs_ori': np.array(cols['imgs_ori']) if use_ori else None}
