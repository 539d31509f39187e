"""DummyReader -- drop-in for tssep/data.py:11-152 (deterministic synthetic 8-speaker data).
Regenerates the reference's examples bit-exactly (tests/golden/dummy_reader.npz)."""
import dataclasses

import numpy as np

from . import dataset as lazy_dataset
from .configurable import Configurable


@dataclasses.dataclass
class DummyReader(Configurable):
    train_dataset_name: str = "train"
    validate_dataset_name: str = "validate"
    domain_adaptation_src_dataset_name: str = "validate"
    eval_dataset_name: str = "eval"
    sample_rate: int = 16000
    aux_size: int = 100
    train_examples: int = 10

    def _get_vad(self, num_samples, num_speakers):           # data.py:34-56
        vad = np.zeros((num_speakers, num_samples), dtype=bool)
        start = 0
        for i in range(num_speakers):
            end = num_samples * (i + 2) // (num_speakers + 1)
            vad[i, start:end] = True
            start = end - (end - start) // 2
        return vad

    def get_example(self, seed, dataset_name, load_keys=("observation",
                                                         "speaker_reverberation_early_ch0", "vad"),
                    num_speakers=8, seconds=5):
        num_samples = self.sample_rate * seconds
        rng = np.random.RandomState(seed)
        max_frequency, min_frequency, num_frequencies = 7000, 100, 3
        frequency = rng.randint(min_frequency, max_frequency, size=(num_frequencies, num_speakers))
        time = np.arange(num_samples) / self.sample_rate
        early = np.sin(2 * np.pi * frequency[..., None] * time).sum(axis=0).astype(np.float32)
        early = early[:, None, :]
        vad = self._get_vad(num_samples, num_speakers)
        early = early * vad[:, None, :]
        noise = 1 * rng.rand(1, num_samples).astype(np.float32)
        observation = early.sum(axis=0) + noise
        aux = np.full((num_speakers, self.aux_size), fill_value=0, dtype=np.float32)
        scale = max_frequency + 1
        for spk, fs in enumerate(frequency.T):
            for f in fs:
                f = (f * aux.shape[1]) // scale
                aux[spk, f:f + 2] = 1
        r = {"example_id": f"dummy_id_{seed}", "num_samples": num_samples,
             "audio_data": {"observation": observation,
                            "speaker_reverberation_early_ch0": early[:, 0], "vad": vad},
             "auxInput": aux, "dataset": dataset_name}
        if "speaker_reverberation_early_ch0" not in load_keys:
            del r["audio_data"]["speaker_reverberation_early_ch0"]
        return r

    def __call__(self, dataset_name, pre_load_apply=None, load_keys=("observation",)):
        n = self.train_examples if "train" in dataset_name else 4
        examples = [self.get_example(i, dataset_name, load_keys) for i in range(n)]
        ds = lazy_dataset.new(examples)                       # data.py:141-144
        if pre_load_apply is not None:
            ds = pre_load_apply(ds)
        return ds

    class data_hooks:
        @staticmethod
        def pre_net(ex):
            return ex
