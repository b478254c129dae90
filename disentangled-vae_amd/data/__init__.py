from .moving_mnist import DeviceClipLoader, MovingMNIST, MovingMNISTWithLabels, get_moving_mnist_dataloaders

__all__ = ['DeviceClipLoader', 'MovingMNIST', 'MovingMNISTWithLabels', 'get_moving_mnist_dataloaders']
