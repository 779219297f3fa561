"""PointCNN part segmentation -- counterpart of /root/reference/networks/seg/pointcnn_partseg.py:13-49.

Encoder: X-conv stages 3->256 (K=8, all points), 256->256 (K=12, 768 FPS points), 256->512 (K=16, 384),
512->1024 (K=16, 128); decoder: four ``RandPointCNN_Decoder`` stages walking back up, each an X-conv from the coarse
level onto the finer level's points, concatenated with that level's encoder features and fused by a ``Dense_Conv1d``
(:24-27, :38-45).  Output ``[B,part_num,N]`` like the reference (:47).
"""
from torch import nn

from ...misc.pointcnn import RandPointCNN, RandPointCNN_Decoder


def EncoderCNN(a, b, c, d, e):
    return RandPointCNN(a, b, 3, c, d, e)


def DecoderCNN(a, b, last_c, c, d, e):
    return RandPointCNN_Decoder(a, b, last_c, 3, c, d, e)


class PointCNN_partseg(nn.Module):
    def __init__(self, part_num=50):
        super().__init__()
        self.encoder_0 = EncoderCNN(3, 256, 8, 1, -1)
        self.encoder_1 = EncoderCNN(256, 256, 12, 1, 768)
        self.encoder_2 = EncoderCNN(256, 512, 16, 1, 384)
        self.encoder_3 = EncoderCNN(512, 1024, 16, 1, 128)
        self.decoder_0 = DecoderCNN(1024, 1024, 1024, 16, 1, 128)
        self.decoder_1 = DecoderCNN(1024, 512, 512, 16, 1, 385)
        self.decoder_2 = DecoderCNN(512, 256, 256, 12, 1, 768)
        self.decoder_3 = DecoderCNN(256, part_num, 256, 8, 1, 2048)

    def forward(self, x, normal=None):
        """x [B,N,3] -> per-point part logits [B,part_num,N]."""
        x = (x, x)
        x_0 = self.encoder_0(x)
        x_1 = self.encoder_1(x_0)
        x_2 = self.encoder_2(x_1)
        x_3 = self.encoder_3(x_2)
        x_3 = self.decoder_0(x_3, x_3)
        x_2 = self.decoder_1(x_3, x_2)
        x_1 = self.decoder_2(x_2, x_1)
        x_0 = self.decoder_3(x_1, x_0)
        return x_0[1].permute(0, 2, 1)

    def execute(self, *a, **k):
        return self(*a, **k)
