import torch


class _FakeResNet(torch.nn.Module):
    """Same attribute names as torchvision ResNet; tiny convs (never executed by the harness)."""

    def __init__(self):
        super().__init__()
        self.conv1 = torch.nn.Conv2d(3, 4, 3)
        self.bn1 = torch.nn.BatchNorm2d(4)
        self.relu = torch.nn.ReLU()
        self.maxpool = torch.nn.MaxPool2d(2)
        self.layer1 = torch.nn.Identity()
        self.layer2 = torch.nn.Identity()
        self.layer3 = torch.nn.Identity()
        self.layer4 = torch.nn.Identity()


def resnet18(pretrained=False):
    return _FakeResNet()


resnet34 = resnet50 = resnet18
