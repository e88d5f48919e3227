"""Random search and regularized evolution over sub-networks of a trained supernet (reference: nasrec/searcher/searcher.py).

Parallelism is the reference's: independent candidates, one process per GPU (`gpu_id = job_id`), no collective — on an 8-GPU
MI355X node `--num_parallel_workers 8` scores 8 candidates at a time.  Workers are spawned (never forked: a process that has
touched the GPU must not be duplicated) and never re-exec."""
import argparse
from copy import deepcopy

import numpy as np
import torch.multiprocessing as mp

from ..search_space import ops_config_lib
from .searcher_utils import create_model_train_and_get_results_helper, get_device_id
from .tokenizer import Tokenizer

_CRITERIA = ["test_loss", "test_acc", "test_auroc", "test_loss_penalty_lat"]


class Searcher(object):
    def __init__(self, eval_fn, args: argparse.Namespace):
        self._eval_fn = eval_fn
        self._args = args
        self.all_results = []
        self._tokenizer = Tokenizer(num_blocks=args.num_blocks, ops_config=ops_config_lib[args.config])
        self._checkpoint = None

    @staticmethod
    def _sort_results_with_criterion(results: np.ndarray, criterion: str = "test_loss", **kwargs) -> np.ndarray:
        """searcher.py:57-79 (latency-penalised objective: loss + beta * (latency / target - 1))"""
        objs = []
        for r in results:
            if criterion == "test_loss_penalty_lat":
                objs.append(r["test_loss"] + kwargs["beta"] * (r["latency"] / kwargs["target_latency"] - 1))
            else:
                objs.append(r[criterion])
        order = np.argsort(np.asarray(objs).flatten())
        return results[order[::-1]] if criterion in ("test_acc", "test_auroc") else results[order]

    def _sampler(self, population: np.ndarray, num_samples: int, criterion: str = "test_loss") -> np.ndarray:
        return population[np.random.choice(len(population), num_samples, replace=False)]

    @staticmethod
    def _defaults(kwargs):
        kwargs["beta"] = kwargs.get("beta", 0.0)
        kwargs["target_latency"] = kwargs.get("target_latency", -1)
        kwargs["latency_batch_size"] = kwargs.get("latency_batch_size", 512)
        if kwargs["target_latency"] == -1 and kwargs["beta"] != 0:
            kwargs["target_latency"] = 0.0
        return kwargs

    def _run_jobs(self, choices, on_cpu, ckpt_holder, kwargs):
        """one process per candidate of this wave (searcher.py:134-152 / 261-278) -> list of result dicts"""
        ctx = mp.get_context("spawn")
        manager = self._manager
        return_dict = manager.dict()
        procs = []
        for job_id, choice in enumerate(choices):
            p = ctx.Process(target=create_model_train_and_get_results_helper,
                            args=(deepcopy(self._args), get_device_id(job_id, on_cpu), self._eval_fn, deepcopy(self._tokenizer), choice,
                                  return_dict, ckpt_holder, kwargs))
            p.start()
            procs.append(p)
        for p in procs:
            p.join()
            if p.exitcode != 0:
                raise RuntimeError("a search worker exited with code %s" % p.exitcode)
        return [return_dict[k] for k in sorted(return_dict.keys())]

    def random_search_from_supernet(self, budget: int = 200, criterion: str = "test_loss", top_k: int = 5, num_parallel_workers: int = 1,
                                    on_cpu: bool = False, sorted: bool = True, **kwargs) -> np.ndarray:
        """searcher.py:88-165"""
        assert num_parallel_workers >= 1, "Should have at least 1 worker!"
        assert top_k <= budget, "Should have 'top_k' smaller than 'budget'."
        if on_cpu:
            assert num_parallel_workers == 1, ValueError("Can only use 'num_parallel_workers=1' when on CPU.")
        assert criterion in _CRITERIA, NotImplementedError("Criterion {} is not supported!".format(criterion))
        kwargs = self._defaults(kwargs)
        self.all_results = []
        self._manager = mp.Manager()
        ckpt_holder = self._manager.dict()
        idx = 0
        while idx < budget:
            print("Evaluating {} of {} random networks!".format(idx, budget))
            num_jobs = min(num_parallel_workers, budget - idx)
            self.all_results += self._run_jobs([None] * num_jobs, on_cpu, ckpt_holder, kwargs)
            idx += num_jobs
        self.all_results = np.asarray(self.all_results)
        if sorted:
            return self._sort_results_with_criterion(self.all_results, criterion, **kwargs)[:top_k]
        return self.all_results[:top_k]

    def regularized_evolution_from_supernet(self, n_generations: int = 50, n_childs: int = 16, init_population: int = 100, sample_size: int = 5,
                                            criterion: str = "test_loss", skip_random: bool = False, num_parallel_workers: int = 1,
                                            on_cpu: bool = False, top_k: int = 2, **kwargs) -> np.ndarray:
        """searcher.py:167-294: aging evolution — tournament of `sample_size`, the winner's mutated children join the population,
        the oldest `n_childs` members leave; the `top_k` best children of every generation form the returned history."""
        assert criterion in _CRITERIA, NotImplementedError("Criterion {} is not supported!".format(criterion))
        assert top_k <= sample_size, ValueError("You must maintain more than 'top_k' children to append 'top_k' archs to history.")
        assert sample_size < init_population, ValueError("Sample size must be no greater than the number of population ('init_population')!")
        assert num_parallel_workers >= 1, ValueError("Should have at least 1 worker!")
        if init_population < n_childs:
            print("WARNING: For the best effect, you should have more initial population than children!")
        if on_cpu:
            assert num_parallel_workers == 1, ValueError("Can only use 'num_parallel_workers=1' when on CPU.")
        kwargs = self._defaults(kwargs)
        population = np.asarray(self.random_search_from_supernet(budget=init_population, criterion=criterion, top_k=init_population,
                                                                 num_parallel_workers=num_parallel_workers, on_cpu=on_cpu, sorted=False, **kwargs))
        print("Done random sample!")
        history, visited = [], []
        ckpt_holder = self._manager.dict()
        for n_gen in range(n_generations):
            parent = self._sort_results_with_criterion(self._sampler(population, sample_size, criterion), criterion, **kwargs)[0]
            print("Parent Arch: {}".format(parent))
            # more mutations early, fewer late; one per child below 20 generations (searcher.py:243)
            num_mutations = (n_generations - n_gen) // (max(20, n_generations // 5)) + 1
            child_id = 0
            while child_id < n_childs:
                num_jobs = min(n_childs - child_id, num_parallel_workers)
                print("Generation {}, Child {}!".format(n_gen, child_id))
                wave = []
                for _ in range(num_jobs):
                    mutated = deepcopy(parent["choice"])
                    while True:
                        for _ in range(num_mutations):
                            mutated = self._tokenizer.mutate_spec(mutated)
                        h = self._tokenizer.hash_token(self._tokenizer.tokenize(mutated))
                        if h not in visited:
                            visited.append(h)
                            break
                    wave.append(mutated)
                for r in self._run_jobs(wave, on_cpu, ckpt_holder, kwargs):
                    population = np.append(population, r)
                child_id += num_parallel_workers
            best = self._sort_results_with_criterion(population[-n_childs:], criterion, **kwargs)
            history += [best[i] for i in range(top_k)]
            population = population[n_childs:]
        return np.asarray(history)
