"""ORACLE (CPU baseline leg) -- test infrastructure, never the product path.

The reference's CPU path restated with the *same stock PyTorch CPU operators* the
reference uses (nn.Linear -> addmm, nn.LSTM -> MKLDNN RNN, torch.stft / torch.istft), so
that `bench.py`'s `cpu_baseline` times what the reference would cost on the GPU box's host
cores (the reference's Python files cannot travel there).  Kind = "port".

Checked against the reference-generated golden vectors in tests/test_oracle_golden.py.
Follows: bsrnn.py:385-443 (forward), :445-510 (forward_recurrent), infer.py:29-37 (sandwich).
"""
import torch
import torch.nn.functional as F

HID = 64


class TorchCpuBSRNN:
    def __init__(self, sd, v):
        """sd: key -> array-like (reference state_dict names); v: band widths incl. the 0 band."""
        self.v = list(v)
        self.p = {k: torch.as_tensor(a, dtype=torch.float32).clone() for k, a in sd.items()}
        self.rnn = []
        for j in range(4):
            bidir = (j % 2 == 0)
            m = torch.nn.LSTM(HID, HID, batch_first=True, num_layers=2, bidirectional=bidir)   # bsrnn.py:70
            own = m.state_dict()
            for k in own:
                own[k] = self.p["lstms.%d.m.rnn.%s" % (j, k)]
            m.load_state_dict(own)
            m.eval()
            self.rnn.append(m)

    def _lin(self, x, prefix):
        return F.linear(x, self.p[prefix + ".weight"], self.p[prefix + ".bias"])

    def _block(self, j, x, state=None):
        """NormRNNResidual (bsrnn.py:78-98) on [N, L, 64]."""
        u = self._lin(x, "lstms.%d.m.fc_in" % j)
        if state is None:
            o, st = self.rnn[j](u)
        else:
            o, st = self.rnn[j](u, state)
        return self._lin(o, "lstms.%d.m.fc" % j) + x, st

    def _dual_path(self, z, state=None):
        C, T, K, _ = z.shape
        new_state = []
        si = 0
        for j in range(4):
            if j % 2 == 0:                                                    # bsrnn.py:138-153
                z = self._block(j, z.reshape(C * T, K, HID))[0].reshape(C, T, K, HID)
            else:                                                             # bsrnn.py:106-128
                x = z.permute(0, 2, 1, 3).reshape(C * K, T, HID)
                st = None
                if state is not None:
                    st = (state[2 * si].contiguous(), state[2 * si + 1].contiguous())
                o, (h, c) = self._block(j, x, st)
                new_state.append(torch.stack((h, c), 0))
                si += 1
                z = o.reshape(C, K, T, HID).permute(0, 2, 1, 3)
        return z, torch.cat(new_state, 0)

    def _front(self, xt):
        act = F.leaky_relu
        residual, feats = [], []
        pos = 0
        for i, w in enumerate(self.v):
            a = 2 * w
            if w == 0:
                residual.append(None)
                feats.append(self.p["bandFCs.%d.0.trainable_constant" % i].expand(xt.shape[0], xt.shape[1], HID))
                continue
            b = xt[:, :, pos:pos + a]
            pos += a
            y = act(self._lin(act(self._lin(b, "bandFCs_pre.%d.0" % i)), "bandFCs_pre.%d.2" % i))
            residual.append(y)
            y = act(self._lin(y, "bandFCs.%d.0" % i))
            y = act(self._lin(y, "bandFCs.%d.2" % i))
            feats.append(self._lin(y, "bandFCs.%d.4" % i))
        return residual, torch.stack(feats, 2)

    def _back(self, z, residual):
        act = F.leaky_relu
        parts = []
        for i, w in enumerate(self.v):
            if w == 0:
                continue
            b = z[:, :, i, :]
            b = act(self._lin(b, "bandFCs_back.%d.0" % i))
            b = act(self._lin(b, "bandFCs_back.%d.2" % i))
            b = act(self._lin(b, "bandFCs_back.%d.4" % i))
            b = self._lin(act(self._lin(b, "bandFCs_back_post.%d.0" % i)), "bandFCs_back_post.%d.2" % i)
            parts.append(residual[i] + b)
        return torch.cat(parts, 2)

    def forward_differentiable(self, x):
        """[C,2050,T] -> [C,2050,T] with the autograd graph kept: the reference of the training-step tests (what
        train.py:97-115 differentiates).  Parameters: `trainable()`."""
        xt = x.permute(0, 2, 1)
        residual, z = self._front(xt)
        z, _ = self._dual_path(z)
        return x * self._back(z, residual).permute(0, 2, 1)

    def trainable(self):
        """name -> leaf tensor with requires_grad, under the reference's state_dict names."""
        out = {}
        for k, t in self.p.items():
            if ".m.rnn." not in k:
                out[k] = t.requires_grad_(True)
        for j, m in enumerate(self.rnn):
            for k, t in m.named_parameters():
                out["lstms.%d.m.rnn.%s" % (j, k)] = t
        return out

    @torch.no_grad()
    def forward(self, x):
        """[C,2050,T] -> [C,2050,T]"""
        return self.forward_differentiable(x)

    @torch.no_grad()
    def forward_recurrent(self, x, state):
        """[C,2050], [4,2,C*K,64] -> (y, new_state)"""
        residual, z = self._front(x[:, None, :])
        z, ns = self._dual_path(z, state)
        return x * self._back(z, residual)[:, 0, :], ns

    @torch.no_grad()
    def separate(self, wave):
        """infer.py:29-37 on [C, n] float32 -> [C, (T-1)*1024]."""
        win = torch.hann_window(2048)
        X = torch.stft(wave, n_fft=2048, hop_length=1024, return_complex=True, window=win)
        x = torch.stack((X.real, X.imag), dim=2)
        x = x.reshape(x.shape[0], x.shape[1] * 2, x.shape[3])
        y = self.forward(x)
        y = y.reshape(y.shape[0], -1, 2, y.shape[2])
        Y = torch.complex(y[:, :, 0, :], y[:, :, 1, :])
        return torch.istft(Y, n_fft=2048, hop_length=1024, window=win)
