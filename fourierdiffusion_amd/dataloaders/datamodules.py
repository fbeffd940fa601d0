"""Datasets / datamodules for the entry points -- the parts of fdiff.dataloaders.datamodules the hot path
depends on (reference: src/fdiff/dataloaders/datamodules.py:23-161, 239-294).

Kept semantics: `DiffusionDataset` applies the DFT ONCE at construction (:42-43), keeps per-(t,c) mean / unbiased std
of the reference split (:52-53) and standardises items on access (:61-62); the validation loader standardises with
TRAIN statistics (:128-142); `dataset_parameters` / `feature_mean_and_std` (:145-161).  Different underneath: the
series stay on the GPU, the DFT and the standardisation run in the engine (one fused kernel), batches are slices of a
device tensor instead of per-item collation.

Real datasets of the reference (ECG, MIMIC-III, NASDAQ, NASA, droughts) need Kaggle / credentialed downloads and are
out of scope; `SyntheticDatamodule` (the reference's sine DGP) and `TensorDatamodule` are provided.
"""
from __future__ import annotations

import logging
import os
from pathlib import Path
from typing import Any, Dict, Iterator, Optional, Tuple

import numpy as np
import torch

from ..utils.dataclasses import DiffusableBatch
from ..utils.fourier import dft


def _device() -> torch.device:
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


class DiffusionDataset:
    def __init__(self, X: torch.Tensor, y: Optional[torch.Tensor] = None, fourier_transform: bool = False,
                 standardize: bool = False, X_ref: Optional[torch.Tensor] = None) -> None:
        dev = _device()
        X = X.to(dev, torch.float32)
        if fourier_transform:
            X = dft(X).detach()
        self.X = X
        self.y = y
        self.standardize = standardize
        if X_ref is None:
            X_ref = X
        else:
            X_ref = X_ref.to(dev, torch.float32)
            if fourier_transform:
                X_ref = dft(X_ref).detach()
        self.feature_mean = X_ref.mean(dim=0)
        self.feature_std = X_ref.std(dim=0)
        self._Xs: Optional[torch.Tensor] = None

    def __len__(self) -> int:
        return len(self.X)

    def standardized(self) -> torch.Tensor:
        """All items, standardised once (host plumbing: elementwise on the resident tensor)."""
        if self._Xs is None:
            self._Xs = ((self.X - self.feature_mean) / self.feature_std) if self.standardize else self.X
        return self._Xs

    def __getitem__(self, index: int) -> Dict[str, torch.Tensor]:
        data = {"X": self.standardized()[index]}
        if self.y is not None:
            data["y"] = self.y[index]
        return data


class BatchLoader:
    """Iterates DiffusableBatch slices of a device-resident dataset (the DataLoader + collate_batch of the
    reference, datamodules.py:109-114).  `rank/world` shard every GLOBAL batch for data-parallel training: rank r takes
    the strided slice idx[r::world].  Every rank yields exactly one item per global batch -- also when its slice is empty
    (a last batch smaller than `world`): the Trainer still joins the gradient all-reduce with a zero contribution, so
    the collective counts of the ranks can never diverge.  Each item carries `global_size` (samples of the global batch)
    so that the exchange can weight a rank by n_local / n_global (unequal slices would otherwise bias the mean)."""

    def __init__(self, dataset: DiffusionDataset, batch_size: int, shuffle: bool, rank: int = 0, world: int = 1,
                 epoch: int = 0) -> None:
        self.dataset, self.batch_size, self.shuffle, self.rank, self.world = dataset, batch_size, shuffle, rank, world
        self.epoch = epoch                                  # advances with every pass over the loader
        self.shuffle_seed = int(torch.initial_seed())      # torch.manual_seed(cfg.random_seed): the same on every rank

    def __len__(self) -> int:
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[DiffusableBatch]:
        n = len(self.dataset)
        Xs = self.dataset.standardized()
        if not self.shuffle:
            order = torch.arange(n)
        else:
            # Data-parallel ranks must cut the SAME permutation.  It comes from a generator of its own, seeded by (seed, epoch)
            # -- like the per-epoch generator of torch's RandomSampler -- so it cannot depend on how much of torch's global
            # generator a rank has consumed (a rank with an empty slice of a small last batch runs no training step:
            # 87 553 % 64 == 1 in the reference's ECG set), and a single process shuffles exactly as N ranks do
            g = torch.Generator(device="cpu").manual_seed((self.shuffle_seed * 1000003 + self.epoch) & ((1 << 62) - 1))
            order = torch.randperm(n, generator=g)
        self.epoch += 1
        for i in range(0, n, self.batch_size):
            idx = order[i:i + self.batch_size]
            n_global = int(idx.numel())
            if self.world > 1:
                idx = idx[self.rank::self.world]
            y = None if self.dataset.y is None else self.dataset.y[idx]
            batch = DiffusableBatch(X=Xs.index_select(0, idx.to(Xs.device)).contiguous(), y=y, timesteps=None)
            batch.global_size = n_global            # plain attribute: the reference's dataclass has no such field
            yield batch


class Datamodule:
    def __init__(self, data_dir: Path | str = Path.cwd() / "data", random_seed: int = 42, batch_size: int = 32,
                 fourier_transform: bool = False, standardize: bool = False) -> None:
        if isinstance(data_dir, str):
            data_dir = Path(data_dir)
        self.data_dir = data_dir / self.dataset_name
        self.random_seed = random_seed
        self.batch_size = batch_size
        self.fourier_transform = fourier_transform
        self.standardize = standardize
        self.X_train = torch.Tensor()
        self.y_train: Optional[torch.Tensor] = None
        self.X_test = torch.Tensor()
        self.y_test: Optional[torch.Tensor] = None
        self.rank, self.world = 0, 1
        self._train_set: Optional[DiffusionDataset] = None
        self._train_src: Optional[torch.Tensor] = None      # the X_train object the cache was built from

    # -- hooks kept from LightningDataModule
    def prepare_data(self) -> None:
        if not self.data_dir.exists():
            logging.info(f"Creating {self.dataset_name} dataset in {self.data_dir}.")
            os.makedirs(self.data_dir)
            self.download_data()

    def download_data(self) -> None:
        raise NotImplementedError

    def setup(self, stage: str = "fit") -> None:
        raise NotImplementedError

    def set_shard(self, rank: int, world: int) -> None:
        self.rank, self.world = rank, world
        self._train_set = None              # rebuilt on the (possibly new) current device

    def invalidate(self) -> None:
        """Call after replacing X_train / y_train: the cached device-resident training set is rebuilt."""
        self._train_set = None

    def _train_dataset(self) -> DiffusionDataset:
        if (self._train_set is None or self._train_set.X.shape[0] != self.X_train.shape[0]
                or self._train_src is not self.X_train):
            self._train_src = self.X_train
            self._train_set = DiffusionDataset(X=self.X_train, y=self.y_train, fourier_transform=self.fourier_transform,
                                               standardize=self.standardize)
        return self._train_set

    def train_dataloader(self) -> BatchLoader:
        epoch = getattr(self, "_train_epoch", 0)           # one loader per epoch (Trainer.fit): the epoch seeds its shuffle
        self._train_epoch = epoch + 1
        return BatchLoader(self._train_dataset(), self.batch_size, shuffle=True, rank=self.rank, world=self.world, epoch=epoch)

    def test_dataloader(self) -> BatchLoader:
        ds = DiffusionDataset(X=self.X_test, y=self.y_test, fourier_transform=self.fourier_transform)
        return BatchLoader(ds, self.batch_size, shuffle=False)

    def val_dataloader(self) -> BatchLoader:
        ds = DiffusionDataset(X=self.X_test, y=self.y_test, fourier_transform=self.fourier_transform,
                              standardize=self.standardize, X_ref=self.X_train)
        return BatchLoader(ds, self.batch_size, shuffle=False)

    @property
    def dataset_name(self) -> str:
        raise NotImplementedError

    @property
    def dataset_parameters(self) -> Dict[str, Any]:
        return {"n_channels": self.X_train.size(2), "max_len": self.X_train.size(1),
                "num_training_steps": (self.X_train.size(0) + self.batch_size - 1) // self.batch_size}

    @property
    def feature_mean_and_std(self) -> Tuple[torch.Tensor, torch.Tensor]:
        ds = self._train_dataset()
        return ds.feature_mean, ds.feature_std


class SyntheticDatamodule(Datamodule):
    """x_n = sin(n f + phi), f ~ Beta(2,2), phi ~ N(0,1) (datamodules.py:285-294), `n_channels` independent draws
    (the reference has one channel; BASELINE's synthetic workloads are named after multi-channel datasets)."""

    def __init__(self, data_dir: Path | str = Path.cwd() / "data", random_seed: int = 42, batch_size: int = 32,
                 fourier_transform: bool = False, standardize: bool = False, max_len: int = 100,
                 num_samples: int = 1000, n_channels: int = 1) -> None:
        super().__init__(data_dir=data_dir, random_seed=random_seed, batch_size=batch_size,
                         fourier_transform=fourier_transform, standardize=standardize)
        self.max_len = max_len
        self.num_samples = num_samples
        self.n_channels = n_channels

    def _file(self) -> Path:
        return self.data_dir / f"sine_T{self.max_len}_C{self.n_channels}_N{self.num_samples}.npy"

    def prepare_data(self) -> None:
        os.makedirs(self.data_dir, exist_ok=True)
        if not self._file().exists():
            self.download_data()

    def setup(self, stage: str = "fit") -> None:
        X = torch.from_numpy(np.load(self._file()))
        self.X_train, self.X_test = X[: self.num_samples], X[self.num_samples:]
        self.y_train = self.y_test = None
        self._train_set = None

    def download_data(self) -> None:
        rng = np.random.RandomState(self.random_seed)
        n = 2 * self.num_samples
        phase = rng.normal(size=(n, 1, self.n_channels))
        frequency = rng.beta(a=2, b=2, size=(n, 1, self.n_channels))
        timesteps = np.arange(self.max_len).reshape(1, -1, 1)
        np.save(self._file(), np.sin(timesteps * frequency + phase).astype(np.float32))

    @property
    def dataset_name(self) -> str:
        return "synthetic"


class TensorDatamodule(Datamodule):
    """Datamodule over in-memory tensors (the role of the reference tests' DummyDatamodule,
    tests/test_datamodules.py:16-53)."""

    def __init__(self, X_train: torch.Tensor, X_test: Optional[torch.Tensor] = None, batch_size: int = 32,
                 fourier_transform: bool = False, standardize: bool = False, random_seed: int = 42) -> None:
        super().__init__(data_dir=Path.cwd(), random_seed=random_seed, batch_size=batch_size,
                         fourier_transform=fourier_transform, standardize=standardize)
        self.X_train = X_train
        self.X_test = X_train if X_test is None else X_test

    def prepare_data(self) -> None:
        pass

    def setup(self, stage: str = "fit") -> None:
        pass

    @property
    def dataset_name(self) -> str:
        return "tensor"


def _needs_download(name: str):
    class _Unavailable(Datamodule):
        def __init__(self, *a, **k):
            raise FileNotFoundError(
                f"{name}: the reference downloads this dataset from Kaggle / PhysioNet "
                "(src/fdiff/dataloaders/datamodules.py); no network here -- use SyntheticDatamodule or TensorDatamodule")
    _Unavailable.__name__ = name
    return _Unavailable


class ECGDatamodule(Datamodule):
    """MIT-BIH heartbeats (datamodules.py:165-238 of the reference): `mitbih_train.csv` / `mitbih_test.csv` under
    `data_dir/ecg`, 187 time steps + a label column.  The two preprocessing options run on the engine:
    `subsample_localization` keeps the 1000 series most localised in time relative to frequency
    (fd_localization_metrics), `smooth_frequency` convolves the spectrum with a Gaussian of width `smoother_width`
    (fd_rfft_pack -> fd_frequency_smooth -> fd_irfft_unpack).  The files themselves come from Kaggle: no network here, so
    a missing directory raises instead of downloading."""

    def __init__(self, data_dir: Path | str = Path.cwd() / "data", random_seed: int = 42, batch_size: int = 32,
                 fourier_transform: bool = False, standardize: bool = False, subsample_localization: bool = False,
                 smooth_frequency: bool = False, smoother_width: float = 0.0) -> None:
        super().__init__(data_dir=data_dir, random_seed=random_seed, batch_size=batch_size,
                         fourier_transform=fourier_transform, standardize=standardize)
        self.subsample_localization = subsample_localization
        self.smooth_frequency = smooth_frequency
        self.smoother_width = smoother_width

    def prepare_data(self) -> None:
        if not (self.data_dir / "mitbih_train.csv").exists():
            self.download_data()

    def download_data(self) -> None:
        raise FileNotFoundError(
            f"ECGDatamodule: {self.data_dir}/mitbih_train.csv not found; the reference downloads shayanfazeli/heartbeat "
            "from Kaggle (datamodules.py:229-235) and there is no network here -- place the two CSV files there")

    def setup(self, stage: str = "fit") -> None:
        import pandas as pd
        from ..utils.fourier import localization_metrics, smooth_frequency
        df_train = pd.read_csv(self.data_dir / "mitbih_train.csv")     # (first row is consumed as the header, as in the
        df_test = pd.read_csv(self.data_dir / "mitbih_test.csv")       #  reference: datamodules.py:196-201)
        self.X_train = torch.tensor(df_train.iloc[:, :187].values, dtype=torch.float32).unsqueeze(2)
        self.y_train = torch.tensor(df_train.iloc[:, 187].values, dtype=torch.long)
        self.X_test = torch.tensor(df_test.iloc[:, :187].values, dtype=torch.float32).unsqueeze(2)
        self.y_test = torch.tensor(df_test.iloc[:, 187].values, dtype=torch.long)
        if self.subsample_localization:                                # datamodules.py:207-219
            X_loc, X_spec_loc = localization_metrics(self.X_train)
            idx_ranking = torch.argsort(X_loc / X_spec_loc, descending=False)
            self.X_train = self.X_train[idx_ranking[:1000]]
            self.y_train = self.y_train[idx_ranking[:1000]]
            X_loc, X_spec_loc = localization_metrics(self.X_train)
            logging.info("Subsampling the training set based on localization metrics.")
            logging.info(f"New time delocalization: {X_loc.mean().item():.3g}")
            logging.info(f"New frequency delocalization: {X_spec_loc.mean().item():.3g}")
        if self.smooth_frequency and self.smoother_width > 0.0:        # datamodules.py:221-230
            self.X_train = smooth_frequency(self.X_train, sigma=self.smoother_width)
            self.X_test = smooth_frequency(self.X_test, sigma=self.smoother_width)
            logging.info("Smoothing the frequency domain of the data.")
            X_loc, X_spec_loc = localization_metrics(self.X_train)
            logging.info(f"New time delocalization: {X_loc.mean().item():.3g}")
            logging.info(f"New frequency delocalization: {X_spec_loc.mean().item():.3g}")

    @property
    def dataset_name(self) -> str:
        return "ecg"


MIMICIIIDatamodule = _needs_download("MIMICIIIDatamodule")
NASDAQDatamodule = _needs_download("NASDAQDatamodule")
NASADatamodule = _needs_download("NASADatamodule")
USDroughtsDatamodule = _needs_download("USDroughtsDatamodule")
