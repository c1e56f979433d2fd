"""Model containers with the reference's constructor signatures and state_dict key
layout (python/models/models.py:41-62, 90-133, 184-197), so checkpoints written by
the reference's training scripts load unchanged (`load_state_dict(torch.load(path,
map_location="cpu"))`, scripts/evaluate_M1.py:189-190).

These modules are parameter holders plus a plain torch forward (used by the plotting
scripts, scripts/reconstruct_M1.py:94-163); the MCEM hot path never calls them: it
takes the weights out of the state_dict and runs the HIP kernels."""
import torch
from torch import nn
from torch.nn import init


class GaussianSample(nn.Module):
    """models.py:24-38: (z, mu, log_var) with z = mu + exp(log_var/2) * randn."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.mu = nn.Linear(in_features, out_features)
        self.log_var = nn.Linear(in_features, out_features)

    def forward(self, x):
        mu, log_var = self.mu(x), self.log_var(x)
        eps = torch.randn(mu.size()).to(mu.device)           # models.py:10 (CPU generator)
        return mu.addcmul(log_var.mul(0.5).exp(), eps), mu, log_var


class Encoder(nn.Module):
    def __init__(self, dims, sample_layer=GaussianSample):
        super().__init__()
        x_dim, h_dim, z_dim = dims
        neurons = [x_dim, *h_dim]
        self.hidden = nn.ModuleList([nn.Linear(neurons[i - 1], neurons[i]) for i in range(1, len(neurons))])
        self.sample = sample_layer(h_dim[-1], z_dim)

    def forward(self, x):
        for layer in self.hidden:
            x = torch.tanh(layer(x))
        return self.sample(x)


class Decoder(nn.Module):
    def __init__(self, dims):
        super().__init__()
        z_dim, h_dim, x_dim = dims
        neurons = [z_dim, *h_dim]
        self.hidden = nn.ModuleList([nn.Linear(neurons[i - 1], neurons[i]) for i in range(1, len(neurons))])
        self.reconstruction = nn.Linear(h_dim[-1], x_dim)

    def forward(self, x):
        for layer in self.hidden:
            x = torch.tanh(layer(x))
        return torch.exp(self.reconstruction(x))


def _xavier(module):
    for m in module.modules():
        if isinstance(m, nn.Linear):
            init.xavier_normal_(m.weight.data)
            if m.bias is not None:
                m.bias.data.zero_()


class VariationalAutoencoder(nn.Module):
    def __init__(self, dims):
        super().__init__()
        x_dim, z_dim, h_dim = dims
        self.z_dim = z_dim
        self.encoder = Encoder([x_dim, h_dim, z_dim])
        self.decoder = Decoder([z_dim, list(reversed(h_dim)), x_dim])
        self.kl_divergence = 0
        _xavier(self)

    def forward(self, x, y=None):
        z, z_mu, z_log_var = self.encoder(x)
        self.kl_divergence = -0.5 * torch.sum(z_log_var - z_mu.pow(2) - z_log_var.exp(), axis=-1)
        return self.decoder(z), z_mu, z_log_var

    def sample(self, z):
        return self.decoder(z)


class DeepGenerativeModel(VariationalAutoencoder):
    def __init__(self, dims, classifier):
        x_dim, self.y_dim, z_dim, h_dim = dims
        super().__init__([x_dim, z_dim, h_dim])
        self.encoder = Encoder([x_dim + self.y_dim, h_dim, z_dim])
        self.decoder = Decoder([z_dim + self.y_dim, list(reversed(h_dim)), x_dim])
        self.classifier = classifier
        _xavier(self)

    def forward(self, x, y):
        z, z_mu, z_log_var = self.encoder(torch.cat([x, y], dim=1))
        return self.decoder(torch.cat([z, y], dim=1)), z_mu, z_log_var

    def classify(self, x):
        return self.classifier(x)

    def sample(self, z, y):
        return self.decoder(torch.cat([z, y.float()], dim=1))


class Classifier(nn.Module):
    def __init__(self, dims, batch_norm=False):
        super().__init__()
        x_dim, h_dim, y_dim = dims
        neurons = [x_dim, *h_dim]
        layers = []
        for i in range(1, len(neurons)):
            layers.append(nn.Linear(neurons[i - 1], neurons[i]))
            if batch_norm:
                layers.append(nn.BatchNorm1d(neurons[i]))
        self.hidden = nn.ModuleList(layers)
        self.output_layer = nn.Linear(h_dim[-1], y_dim)

    def forward(self, x):
        for layer in self.hidden:
            x = torch.relu(layer(x))
        return torch.sigmoid(self.output_layer(x))


class Classifier2Classes(nn.Module):
    """models.py:64-88: the Classifier with 2 * y_dim outputs and a softmax over the two classes; forward returns
    (N, 2, y_dim).  (vaenmf.engine.classifier_layers_from_state(sd, two_classes=True) gives the HIP path's layers.)"""

    def __init__(self, dims, batch_norm=False):
        super().__init__()
        x_dim, h_dim, y_dim = dims
        neurons = [x_dim, *h_dim]
        layers = []
        for i in range(1, len(neurons)):
            layers.append(nn.Linear(neurons[i - 1], neurons[i]))
            if batch_norm:
                layers.append(nn.BatchNorm1d(neurons[i]))
        self.hidden = nn.ModuleList(layers)
        self.output_layer = nn.Linear(h_dim[-1], 2 * y_dim)
        self.softmax = nn.Softmax(dim=1)
        self.y_dim = y_dim

    def forward(self, x):
        for layer in self.hidden:
            x = torch.relu(layer(x))
        return self.softmax(self.output_layer(x).view(-1, 2, self.y_dim))
