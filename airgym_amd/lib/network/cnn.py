"""Depth-image feature extractor (reference: lib/network/cnn.py:3-33): three stride-2 convolutions, each
followed by ReLU and BatchNorm, global average pooling, one Linear to `feature_dim`.  Module names
(`features.<i>`, `fc`) are the reference's so that `trained/planning_cnn_rate.pth` style checkpoints load."""
import torch.nn as nn


class CNNFeatureExtractor(nn.Module):
    def __init__(self, feature_dim=12):
        super().__init__()
        layers = []
        for cin, cout, k, pad in ((1, 16, 5, 2), (16, 32, 3, 1), (32, 64, 3, 1)):   # (1,212,120) -> (64,27,15)
            layers += [nn.Conv2d(cin, cout, kernel_size=k, stride=2, padding=pad), nn.ReLU(), nn.BatchNorm2d(cout)]
        layers.append(nn.AdaptiveAvgPool2d((1, 1)))
        self.features = nn.Sequential(*layers)
        self.fc = nn.Linear(64, feature_dim)

    def forward(self, x):
        x = self.features(x)
        return self.fc(x.view(x.size(0), -1))
