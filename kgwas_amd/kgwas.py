"""``KGWAS`` trainer -- the reference's driver API (kgwas/kgwas.py:25-212) on the MI355X path.

Kept verbatim from the reference: constructor / ``initialize_model`` / ``train`` / ``load_pretrained``
signatures and defaults, the attributes they set (``model``, ``best_model``, ``config``,
``train_loader`` ... ``infer_loader``, ``save_name``, ``kgwas_res``), the LD-weighted MSE
(kgwas.py:142-145), Adam(lr, weight_decay as L2) (kgwas.py:116), best-model selection by validation
Pearson (kgwas.py:170-173), checkpoint files (utils.py:203-207).

Deliberately different (Appendix A of SURVEY.md): no CUDA_LAUNCH_BLOCKING, no per-seed ``.item()``
dict look-ups (LD weights are a resident float64 vector indexed by SNP id), evaluation under no_grad.
Multi-GPU: one process per GPU (torch.distributed, RCCL); every rank trains on its own slice of each
batch and parameter gradients are all-reduced in one flat bucket (kgwas_amd/dist.py).
"""
from __future__ import annotations

import os
import pickle
from copy import deepcopy

import numpy as np
import torch
import torch.nn.functional as F

from . import dist as kdist
from . import ops
from .model import HeteroGNN
from .sampler import NeighborLoader
from .utils import (compute_metrics, evaluate_minibatch_clean, get_network_weight, load_pretrained, print_sys, write_tsv,
                    save_model)


class KGWAS:
    def __init__(self, data, weight_bias_track=False, device='cuda', proj_name='KGWAS', exp_name='KGWAS',
                 seed=42):
        torch.manual_seed(seed)                                     # kgwas.py:33-35
        if torch.cuda.is_available():
            torch.cuda.manual_seed(seed)
        np.random.seed(seed)
        self.seed = seed
        self.device = device if torch.cuda.is_available() else 'cpu'
        self.data = data
        self.data_path = data.data_path
        if weight_bias_track:
            import wandb
            wandb.init(project=proj_name, name=exp_name)
            self.wandb = wandb
        else:
            self.wandb = False
        self.exp_name = exp_name

    def initialize_model(self, gnn_num_layers=2, gnn_hidden_dim=128, gnn_backbone='GAT', gnn_aggr='sum',
                         gat_num_head=1, no_relu=False):
        self.config = {'gnn_num_layers': gnn_num_layers, 'gnn_hidden_dim': gnn_hidden_dim,
                       'gnn_backbone': gnn_backbone, 'gnn_aggr': gnn_aggr, 'gat_num_head': gat_num_head}
        self.gnn_num_layers = gnn_num_layers
        self.model = HeteroGNN(self.data.data, gnn_hidden_dim, 1, gnn_num_layers, gnn_backbone, gnn_aggr,
                               self.data.snp_init_dim_size, self.data.gene_init_dim_size,
                               self.data.go_init_dim_size, gat_num_head, no_relu=no_relu).to(self.device)

    def load_pretrained(self, path):
        import pandas as pd
        with open(os.path.join(path, 'config.pkl'), 'rb') as f:
            config = pickle.load(f)
        self.initialize_model(**config)
        self.config = config
        self.model = load_pretrained(path, self.model).to(self.device)
        self.best_model = self.model
        pred_csv = os.path.join(path, 'pred.csv')
        if os.path.exists(pred_csv):
            self.kgwas_res = pd.read_csv(pred_csv, sep=None, engine='python')
        self.save_name = path.split('/')[-1]

    # ------------------------------------------------------------------------------------------
    def _ld_weight_vector(self) -> torch.Tensor:
        """float64 [N_SNP] LD-score regression weight per SNP index (kgwas.py:142-143 without the
        per-step Python look-ups); SNPs without a weight get 0 and are never seeds."""
        n = int(self.data.data['SNP'].x.shape[0])
        w = torch.zeros(n, dtype=torch.float64)
        ids = np.asarray(self.data.all_ids, dtype=np.int64)
        w[torch.from_numpy(ids)] = torch.from_numpy(np.asarray(self.data.ldsc_weight, dtype=np.float64))
        return w.to(self.device)

    def make_loaders(self, batch_size, num_workers=0):
        kwargs = {'batch_size': batch_size, 'num_workers': num_workers, 'drop_last': True, 'device': self.device}
        eval_kwargs = {'batch_size': 512, 'num_workers': num_workers, 'drop_last': False, 'device': self.device}
        L = self.gnn_num_layers
        rank, world = kdist.rank_world()
        tr_type, tr_ids = self.data.train_input_nodes
        tr_ids = kdist.shard_batches(np.asarray(tr_ids), batch_size, rank, world)
        tr_kwargs = dict(kwargs, batch_size=batch_size // world)      # each rank: its slice of every batch
        self.train_loader = NeighborLoader(self.data.data, num_neighbors=[-1] * L, sampler=None,
                                           input_nodes=(tr_type, tr_ids), **tr_kwargs)        # kgwas.py:99-101
        self.val_loader = NeighborLoader(self.data.data, num_neighbors=[-1] * L,
                                         input_nodes=self.data.val_input_nodes, **kwargs)     # :102-103
        self.test_loader = NeighborLoader(self.data.data, num_neighbors=[-1] * L,
                                          input_nodes=self.data.test_input_nodes, **eval_kwargs)  # :104-105
        infer_idx = np.asarray(self.data.all_ids)                                             # :107-110
        self.infer_loader = NeighborLoader(self.data.data, num_neighbors=[-1] * L,
                                           input_nodes=('SNP', infer_idx), **eval_kwargs)     # :112-113

    def train_step(self, batch, optimizer, ld_w, world: int = 1):
        """kgwas.py:130-151 for one batch; returns the (float64) loss tensor, no host sync."""
        optimizer.zero_grad(set_to_none=True)
        bs = batch['SNP'].batch_size
        # forward + mean(ld_weight * (pred - y)**2) in float64 (kgwas.py:137-145); labels / weights by the seeds' ids
        loss, _ = self.model.forward_loss(batch.x_dict, batch.edge_index_dict, bs, batch.n_id('SNP'), batch.dg.y['SNP'], ld_w,
                                          unit_grad=True)
        loss.backward(gradient=ops.unit_gradient(loss.device))      # (the resident 1.0: no ones_like fill, see readout_weighted_mse)
        if world > 1:
            kdist.allreduce_grads(self.model, world)
        optimizer.step()
        return loss

    def _train_sharded(self, batch_size, lr, weight_decay, total_epoch, save_best_model, save_name, use_graph=True):
        """kgwas/kgwas.py:85-212 in the SNP-sharded multi-GPU mode (kgwas_amd/shard.py): every rank works on the SAME
        batches of the reference's order and owns the SNPs of one id range; validation / test / inference predictions are
        computed shard-wise and summed, so every rank sees the same metrics and keeps the same best model."""
        from .shard import ShardedTrainer
        rank, world = kdist.rank_world()
        if world > 1:
            kdist.broadcast_params(self.model)
        st = ShardedTrainer(self, self.data.train_input_nodes, batch_size, lr=lr, weight_decay=weight_decay, use_graph=use_graph)
        y_all = self.data.data['SNP'].y

        def evaluate(ids, model, drop_last=False):
            ids = np.asarray(ids)
            if drop_last:                                   # the reference's val loader drops the partial batch (kgwas.py:102-103)
                ids = ids[:len(ids) // batch_size * batch_size]
            pred = st.predict(ids, model).cpu().numpy()
            return {'pred': pred, 'truth': y_all[torch.from_numpy(ids)].numpy()}

        min_val = -1000
        self.best_model = deepcopy(self.model).to(self.device)
        print_sys('Start Training (SNP-sharded over %d ranks)...' % world)
        for ep in range(total_epoch):
            self.model.train()
            for step in range(st.n_batches):
                st.step(step)
                if (step % 500 == 0) and (step >= 500):
                    print_sys('Epoch {} Step {} Train Loss (this rank\'s share): {:.4f}'.format(ep + 1, step + 1, float(st.last_loss)))
            st.check()                                      # (captured form: no batch overflowed the static layout)
            val_metrics = compute_metrics(evaluate(self.data.val_input_nodes[1], self.model, True), False, -1, -1, F.mse_loss)
            print_sys('Epoch {}: Validation MSE: {:.4f} Validation Pearson: {:.4f}. '.format(
                ep + 1, val_metrics['mse'], val_metrics['pearsonr']))
            self.val_metrics = val_metrics
            if val_metrics['pearsonr'] > min_val:                    # kgwas.py:170-173
                min_val = val_metrics['pearsonr']
                self.best_model = deepcopy(self.model)
        if save_best_model and rank == 0:
            save_model_path = os.path.join(self.data_path, 'model')
            save_model(self.best_model, self.config, os.path.join(save_model_path, save_name))
        self.test_metrics = compute_metrics(evaluate(self.data.test_input_nodes[1], self.best_model), False, -1, -1, F.mse_loss)
        self.data.lr_uni['pred'] = evaluate(self.data.all_ids, self.best_model)['pred']                  # kgwas.py:189-191
        self._postprocess(save_name, save_best_model and rank == 0)
        self.shard_bytes_moved = st.xchg.bytes_moved

    def train(self, batch_size=512, num_workers=0, lr=1e-4, weight_decay=5e-4, epoch=10, save_best_model=True,
              save_name=None, data_to_cuda=False, use_graph=True, parallelism='seed'):
        """Same signature and defaults as kgwas/kgwas.py:85-87.  ``use_graph`` (extra, default on): run the
        training step as one captured HIP graph (kgwas_amd/graph_step.py); off = eager launches, same math.
        ``parallelism`` (extra; matters under torch.distributed): 'seed' = every rank trains on its slice of each batch
        against a replicated graph (kgwas_amd/dist.py); 'shard' = SNP rows sharded by id range, Gene / GO replicated
        (kgwas_amd/shard.py, BASELINE.json north_star)."""
        total_epoch = epoch
        if save_name is None:
            save_name = self.exp_name
        self.save_name = save_name
        if parallelism == 'shard':
            return self._train_sharded(batch_size, lr, weight_decay, total_epoch, save_best_model, save_name, use_graph)
        if parallelism != 'seed':
            raise ValueError(f"parallelism {parallelism!r}: 'seed' or 'shard'")
        print_sys('Creating data loader...')
        self.make_loaders(batch_size, num_workers)
        rank, world = kdist.rank_world()
        if world > 1:
            kdist.broadcast_params(self.model)
        graph_step = None
        if use_graph and len(self.train_loader) > 0:
            from .graph_step import GraphTrainStep
            # (multi-GPU: a 512-seed batch split over the ranks = strong scaling; from 4 ranks on the batch-independent first gene
            #  Linear is split by gene rows instead of repeated on every rank -- ops.GeneLayerShard)
            graph_step = GraphTrainStep(self, (self.train_loader.input_type, self.train_loader.ids.cpu().numpy()),
                                        self.train_loader.batch_size, lr=lr, weight_decay=weight_decay,
                                        shard_gene_layer=None,
                                        # (the loader's batch order is fixed, kgwas.py:93-101: the batches sampled in epoch 1 are
                                        #  kept in HBM and put back in later epochs -- graph_step.BatchCache)
                                        cache_batches=total_epoch > 1)
            optimizer = graph_step.opt
        else:
            optimizer = torch.optim.Adam(self.model.parameters(), lr=lr, weight_decay=weight_decay)   # kgwas.py:116
        ld_w = self._ld_weight_vector()
        min_val = -1000
        self.best_model = deepcopy(self.model).to(self.device)
        print_sys('Start Training...')
        for ep in range(total_epoch):
            self.model.train()
            steps = range(graph_step.n_batches) if graph_step is not None else enumerate(self.train_loader)
            for item in steps:
                if graph_step is not None:
                    step, loss = item, graph_step.step(item)
                    if step % 128 == 127:                    # a batch that outgrew the static layout: stop soon, not at the end
                        graph_step.poll()                    # of the epoch (no stream sync: the answer is read one poll later)
                else:
                    step, batch = item
                    loss = self.train_step(batch, optimizer, ld_w, world)
                if self.wandb:
                    self.wandb.log({'training_loss': loss.item()})
                if (step % 500 == 0) and (step >= 500):
                    print_sys('Epoch {} Step {} Train Loss: {:.4f}'.format(ep + 1, step + 1, loss.item()))
            if graph_step is not None:
                graph_step.check()
            val_res = evaluate_minibatch_clean(self.val_loader, self.model, self.device)
            val_metrics = compute_metrics(val_res, False, -1, -1, F.mse_loss)
            print_sys('Epoch {}: Validation MSE: {:.4f} Validation Pearson: {:.4f}. '.format(
                ep + 1, val_metrics['mse'], val_metrics['pearsonr']))
            self.val_metrics = val_metrics
            if self.wandb:
                for i, j in val_metrics.items():
                    self.wandb.log({'val_' + i: j})
            if val_metrics['pearsonr'] > min_val:                    # kgwas.py:170-173
                min_val = val_metrics['pearsonr']
                self.best_model = deepcopy(self.model)
        if save_best_model and rank == 0:
            save_model_path = os.path.join(self.data_path, 'model')
            print_sys('Saving models to ' + os.path.join(save_model_path, save_name))
            save_model(self.best_model, self.config, os.path.join(save_model_path, save_name))
        test_res = evaluate_minibatch_clean(self.test_loader, self.best_model, self.device)
        self.test_metrics = compute_metrics(test_res, False, -1, -1, F.mse_loss)
        if self.wandb:
            for i, j in self.test_metrics.items():
                self.wandb.log({'test_' + i: j})
        infer_res = evaluate_minibatch_clean(self.infer_loader, self.best_model, self.device)
        self.data.lr_uni['pred'] = infer_res['pred']                 # kgwas.py:191
        self._postprocess(save_name, save_best_model and rank == 0)

    def get_network_weight(self):
        """The graph-side half of kgwas/kgwas.py:268-273: per-edge raw attention weights of the best model."""
        return get_network_weight(self, self.data)

    def get_disease_critical_network(self, variant_threshold=5e-8, magma_path=None, magma_threshold=0.05,
                                     program_threshold=0.05, K_neighbors=3, num_cpus=1):
        """kgwas/kgwas.py:268-273: (attention table, variant interpretation, disease-critical network).  The attention pass
        runs on the fused kernels (get_network_weight), the tables are host work (utils.generate_viz); the MAGMA / GSEA filter
        (``magma_path``) is not built."""
        from .utils import generate_viz
        df_network_weight = get_network_weight(self, self.data)
        df_variant_interpretation, disease_critical_network = generate_viz(
            self, df_network_weight, self.data_path, variant_threshold, magma_path, magma_threshold, program_threshold,
            K_neighbors, num_cpus)
        return df_network_weight, df_variant_interpretation, disease_critical_network

    def _postprocess(self, save_name, save_best_model):
        """kgwas.py:192-212: prediction-weighted p-values (Storey-Tibshirani pi0 per prediction-quantile bin,
        500 bins), bisection calibration, clip to [0,1], CSVs."""
        import numpy as np
        from .eval_utils import find_closest_x, storey_ribshirani_integrate
        lr_uni_to_save = deepcopy(self.data.lr_uni)
        self.data.lr_uni['abs_pred'] = np.abs(self.data.lr_uni['pred'])
        self.data.lr_uni['SR_P_val'] = storey_ribshirani_integrate(self.data.lr_uni, column='abs_pred', num_bins=500)
        self.data.lr_uni['SR'] = -(np.log10(self.data.lr_uni['SR_P_val'].astype(float).values))
        lr_uni_to_save['P_weighted'] = self.data.lr_uni['SR_P_val']
        scale_factor = find_closest_x(lr_uni_to_save)
        lr_uni_to_save['KGWAS_P'] = (scale_factor * lr_uni_to_save['P_weighted']).clip(lower=0, upper=1)
        out_dir = os.path.join(self.data_path, 'model_pred', 'new_experiments')
        self.kgwas_res = lr_uni_to_save
        if kdist.rank_world()[0] != 0:          # multi-GPU: every rank holds the results, rank 0 alone writes the files
            return
        try:
            os.makedirs(out_dir, exist_ok=True)
            write_tsv(lr_uni_to_save, os.path.join(out_dir, save_name + '_pred.csv'))      # (= to_csv(index=False, sep='\t'), same bytes)
            print('KGWAS prediction and p-values saved to ' + os.path.join(out_dir, save_name + '_pred.csv'))
            if save_best_model:
                write_tsv(lr_uni_to_save, os.path.join(self.data_path, 'model', save_name, 'pred.csv'))
        except OSError as e:   # read-only data_path: keep results in memory
            print_sys(f'could not write predictions: {e}')
