"""Per-frame conditioning networks (torch modules, autograd-capable) with the reference's state_dict keys.

Reference: run_nerf_helpers.py:165-178 (AudioNet_W2L), :182-193 (ExpressionEnc), :210-240 (AudioAttNet),
:21-70 (Embedder / get_embedder); run_nerf_com_trainExpLater.py:28-111 (encode_signal*), :182-204
(rot_to_euler / pose_to_euler_trans).  These run once per frame (~0.35 MFLOP) and only produce the
96 + 42 floats that dfn_fold_bias folds into the decoder's bias vectors."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _mlp(dims, slope=0.02):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            layers.append(nn.LeakyReLU(slope, True))
    return nn.Sequential(*layers)


class AudioNet_W2L(nn.Module):
    """512 -> 256 -> 128 -> 64 (keys encoder.{0,2,4})."""

    def __init__(self):
        super().__init__()
        self.encoder = _mlp([512, 256, 128, 64])

    def forward(self, x):
        return self.encoder(x)


class ExpressionEnc(nn.Module):
    """64 -> 32 -> 32 (keys encoder.{0,2})."""

    def __init__(self):
        super().__init__()
        self.encoder = _mlp([64, 32, 32])

    def forward(self, x):
        return self.encoder(x)


class AudioAttNet(nn.Module):
    """Attention over a window of seq_len frames (keys attentionConvNet.{0,2,4,6,8}, attentionNet.0)."""

    def __init__(self, dim_aud=32, seq_len=8):
        super().__init__()
        self.seq_len, self.dim_aud = seq_len, dim_aud
        chans = [dim_aud, 16, 8, 4, 2, 1]
        conv = []
        for a, b in zip(chans[:-1], chans[1:]):
            conv += [nn.Conv1d(a, b, kernel_size=3, stride=1, padding=1, bias=True), nn.LeakyReLU(0.02, True)]
        self.attentionConvNet = nn.Sequential(*conv)
        self.attentionNet = nn.Sequential(nn.Linear(seq_len, seq_len, bias=True), nn.Softmax(dim=1))

    def forward(self, x):
        y = x[..., :self.dim_aud].permute(1, 0).unsqueeze(0)
        y = self.attentionConvNet(y)
        a = self.attentionNet(y.view(1, self.seq_len)).view(self.seq_len, 1)
        return torch.sum(a * x, dim=0)


class Embedder:
    """NeRF-style embedder; get_embedder(3, 0) -> x, sin/cos(x), sin/cos(2x), sin/cos(4x): 21 dims."""

    def __init__(self, input_dims, include_input, max_freq_log2, num_freqs, log_sampling=True):
        if log_sampling:
            self.freqs = 2. ** torch.linspace(0., max_freq_log2, steps=num_freqs)
        else:
            self.freqs = torch.linspace(2. ** 0., 2. ** max_freq_log2, steps=num_freqs)
        self.include_input = include_input
        self.out_dim = input_dims * (int(include_input) + 2 * num_freqs)

    def embed(self, x):
        out = [x] if self.include_input else []
        for f in self.freqs:
            out += [torch.sin(x * f), torch.cos(x * f)]
        return torch.cat(out, -1)


def get_embedder(multires, i=0):
    if i == -1:
        return nn.Identity(), 3
    e = Embedder(3, True, multires - 1, multires)
    return e.embed, e.out_dim


def rot_to_euler(R):
    """[N,3,3+] -> euler [N,3] on R's own device (the reference hard-codes .cuda(), MAIN:184)."""
    return torch.stack([torch.atan2(R[:, 2, 2], R[:, 1, 2]), torch.asin(-R[:, 0, 2]),
                        torch.atan2(R[:, 0, 0], -R[:, 0, 1])], 1)


def pose_to_euler_trans(poses):
    return torch.cat((rot_to_euler(poses), poses[:, :3, 3]), dim=1)


def _window(x, img_i, half, length):
    left, right = img_i - half, img_i + half
    pad_l, pad_r = max(0, -left), max(0, right - length)
    win = x[max(left, 0):min(right, length)]
    if pad_l:
        win = torch.cat((torch.zeros_like(win)[:pad_l], win), 0)
    if pad_r:
        win = torch.cat((win, torch.zeros_like(win)[:pad_r]), 0)
    return win


def encode_signal(dataset, itr_obj, img_i, dim_aud, AudNet, ExpNet, AudAttNet, global_step, args, len_auds,
                  embed_fn=None):
    """MAIN:28-75.  Returns [aud [1,96], None] for object 0, [None, exp] otherwise."""
    if itr_obj != 0:
        return [None, dataset[itr_obj]['exp'][img_i:img_i + 1]]
    auds, exps = dataset[itr_obj]['auds'], dataset[itr_obj]['exp']
    if global_step >= args.nosmo_iters:
        half = int(args.smo_size / 2)
        a = AudNet(_window(auds, img_i, half, len_auds))
        e = ExpNet(_window(exps, img_i, half, len_auds))
        aud = AudAttNet(torch.cat([a, e], 1)).unsqueeze(0)
    else:
        aud = torch.cat([AudNet(auds[img_i:img_i + 1]), ExpNet(exps[img_i:img_i + 1])], 1)
    return [aud, None]


def encode_signal_torso(dataset, itr_obj, img_i, PoseAttNet, global_step, args, len_poses, embed_fn=None):
    """MAIN:78-111.  [1,42] before nosmo_iters, [42] after (shape quirk kept)."""
    poses = dataset[itr_obj]['poses']
    if global_step >= args.nosmo_iters:
        half = int(args.smo_torse_size / 2)
        left, right = max(img_i - half, 0), min(img_i + half, len_poses)
        et = pose_to_euler_trans(poses[left:right])
        pad_l, pad_r = max(0, half - img_i), max(0, img_i + half - len_poses)
        if pad_l:
            et = torch.cat((torch.zeros_like(et)[:pad_l], et), 0)
        if pad_r:
            et = torch.cat((et, torch.zeros_like(et)[:pad_r]), 0)
        emb = torch.cat((embed_fn(et[:, :3]), embed_fn(et[:, 3:])), 1)
        return PoseAttNet(emb)
    et = pose_to_euler_trans(poses[img_i].unsqueeze(0))
    return torch.cat((embed_fn(et[:, :3]), embed_fn(et[:, 3:])), 1)


def _att_batch(att, windows):
    """AudioAttNet on a batch of windows [B, S, D] -> [B, D] (same arithmetic as AudioAttNet.forward per window)."""
    B, S, D = windows.shape
    y = att.attentionConvNet(windows[..., :att.dim_aud].permute(0, 2, 1))           # [B,1,S]
    a = att.attentionNet(y.reshape(B, S))                                          # [B,S], softmax over the window
    return torch.sum(a.unsqueeze(-1) * windows, dim=1)


def encode_signals_batch(dataset, itr_obj, frame_ids, AudNet, ExpNet, AudAttNet, PoseAttNet, global_step, args,
                         length, embed_fn):
    """encode_signal + encode_signal_torso (MAIN:28-111) for a LIST of frames of object 0 in one pass (SURVEY.md
    8(f) rank 3): every network runs once on all rows instead of once per frame.  Returns (sig [B,96], sig_torso
    [B,42]); row b equals encode_signal(...)[0][0] / encode_signal_torso(...).reshape(-1) of frame_ids[b] up to
    the rounding of batched vs single-row GEMMs."""
    d = dataset[itr_obj]
    auds, exps, poses = d['auds'], d['exp'], d['poses']
    idx = torch.as_tensor(list(frame_ids), dtype=torch.long, device=auds.device)
    et = pose_to_euler_trans(poses[:length])
    if global_step >= args.nosmo_iters:
        def windows(x, half):                      # zero-padded rows at both ends, like _window
            z = x.new_zeros(half, x.shape[1])
            xp = torch.cat((z, x[:length], z), 0)
            return xp.unfold(0, 2 * half, 1)[idx].permute(0, 2, 1)                 # [B, 2*half, D]
        h = int(args.smo_size / 2)
        wa, we = windows(auds, h), windows(exps, h)
        B, S = wa.shape[0], wa.shape[1]
        feat = torch.cat((AudNet(wa.reshape(B * S, -1)), ExpNet(we.reshape(B * S, -1))), 1).reshape(B, S, -1)
        sig = _att_batch(AudAttNet, feat)
        ht = int(args.smo_torse_size / 2)
        wt = windows(et, ht)
        Bt, St = wt.shape[0], wt.shape[1]
        flat = wt.reshape(Bt * St, 6)
        emb = torch.cat((embed_fn(flat[:, :3]), embed_fn(flat[:, 3:])), 1).reshape(Bt, St, -1)
        sigt = _att_batch(PoseAttNet, emb)
    else:
        sig = torch.cat((AudNet(auds[idx]), ExpNet(exps[idx])), 1)
        e = et[idx]
        sigt = torch.cat((embed_fn(e[:, :3]), embed_fn(e[:, 3:])), 1)
    return sig, sigt
