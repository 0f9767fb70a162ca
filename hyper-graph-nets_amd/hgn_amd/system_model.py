"""System models with the reference's class API (src/model/abstract_system_model.py, flag.py, cylinder.py, plate.py): frame ->
graph features on the device, remote-graph expansion, training / validation step, one-step update and rollout around
the MI355X message-passing model.

Per frame the reference runs ~40 small torch ops plus Python loops; here ``build_graph`` is a fixed handful of HIP
launches (include/hgn_features.h): node features, relative edge features (+ edge lengths), one segment max/min pass
for the node dynamics, and three launches per normaliser.  The mesh topology (cells -> two-way edges, receiver CSR) is
computed once per ``cells`` tensor and reused for the whole trajectory (the reference recomputes it every frame,
flag.py:76-78).
"""
import math
from typing import Dict, Tuple

import torch
from torch import nn, Tensor

from . import features, ops, topology
from . import rmp as _rmp
from .modules import MeshGraphNet
from .normalizer import Normalizer
from .util import EdgeSet, MultiGraph, MultiGraphWithPos, NodeType, device


def _masked_mse(target: Tensor, output: Tensor, mask: Tensor) -> Tensor:
    """MSELoss()(target[mask], output[mask]) (flag.py:150-152) without the boolean-index host sync."""
    m = mask.to(output.dtype).unsqueeze(1)
    return (((output - target) ** 2) * m).sum() / (m.sum() * output.shape[1])


class AbstractSystemModel(nn.Module):
    """The reference's system-model base (abstract_system_model.py:10-190) together with the constructor logic its three
    subclasses repeat (flag.py:21-63, cylinder.py:21-63, plate.py:21-67): normalisers, the optional graph balancer and remote
    message passing stages, and the learned MeshGraphNet over the resulting edge-set names.  Attribute names that the
    reference's trainer / pickles touch (`learned_model`, `_*_normalizer`, `_rmp`, `_balancer`, `_remote_graph`,
    `_graph_balancer`, `message_passing_steps`, ...) are kept."""
    _model_type = None

    def __init__(self, params, node_size: int, edge_size: int, remote_edge_size: int = 7,
                 edge_sets=('mesh_edges',)) -> None:
        super().__init__()
        self._params = params
        self.loss_fn = torch.nn.MSELoss()
        for attr, (width, name) in {'_output_normalizer': (3, 'output_normalizer'),
                                    '_node_normalizer': (node_size, 'node_normalizer'),
                                    '_node_dynamic_normalizer': (1, 'node_dynamic_normalizer'),
                                    '_mesh_edge_normalizer': (edge_size, 'mesh_edge_normalizer'),
                                    '_intra_edge_normalizer': (remote_edge_size, 'intra_edge_normalizer'),
                                    '_inter_edge_normalizer': (remote_edge_size, 'inter_edge_normalizer'),
                                    '_hyper_node_normalizer': (3, 'hyper_node_normalizer')}.items():
            setattr(self, attr, Normalizer(size=width, name=name))
        remote_cfg, balance_cfg = params.get('rmp'), params.get('graph_balancer')
        connector = remote_cfg.get('connector')
        # a stage is on unless its YAML entry says 'none' (flag.py:30-35)
        self._rmp = 'none' not in (remote_cfg.get('clustering'), connector)
        self._balancer = balance_cfg.get('algorithm') != 'none'
        self._multi = self._rmp and connector == 'multigraph'
        self._architecture = self._select_architecture(connector)
        self._rmp_frequency = remote_cfg.get('frequency')
        self._balance_frequency = balance_cfg.get('frequency')
        self.message_passing_steps = params.get('message_passing_steps')
        self.message_passing_aggregator = params.get('aggregation')
        self._visualized = False
        self.replay_rollout = True          # forward() without gradients replays a captured HIP graph per topology (graphs.GraphedForwardCache)
        self._fwd_cache = None
        self._edge_sets = list(edge_sets)
        if self._balancer:
            from . import graph_balancer as _gb
            self._graph_balancer = _gb.get_balancer(params)
            self._edge_sets.append('balance')
        if self._rmp:
            self._remote_graph = _rmp.get_rmp(params)
            self._edge_sets.extend(self._remote_graph.initialize(
                self._intra_edge_normalizer, self._inter_edge_normalizer, self._hyper_node_normalizer))
        self.learned_model = MeshGraphNet(
            output_size=params.get('size'), latent_size=128, num_layers=2,
            message_passing_steps=self.message_passing_steps,
            message_passing_aggregator=self.message_passing_aggregator,
            architecture=self._architecture, edge_sets=self._edge_sets).to(device)
        self._cells_key = None
        self._cells_edges = None

    def __getstate__(self):
        """The reference pickles the whole system model at every checkpoint (MeshSimulator.py:492-493 `pickle.dump(self)`, with
        `_network` inside) and deep-copies it for evaluation.  Captured HIP graphs (the rollout replay cache) and the per-mesh
        edge cache are run-time state, not model state: a copy starts without them and captures again on its own device."""
        state = super().__getstate__()
        state['_fwd_cache'] = None
        state['_cells_key'] = None
        state['_cells_edges'] = None
        return state

    def _select_architecture(self, connector):
        return connector if self._rmp else 'none'                     # flag.py:36, cylinder.py:36

    # ---- topology, once per mesh ------------------------------------------------------------------------------
    def _mesh_edges(self, cells: Tensor, deform: bool = False):
        """util.triangles_to_edges on the device, cached while the same ``cells`` tensor (or equal content) comes in."""
        hit = self._cells_key is not None and (
            self._cells_key is cells or (self._cells_key.shape == cells.shape and self._cells_key.device == cells.device
                                         and torch.equal(self._cells_key, cells)))
        if not hit:
            s, r, _ = features.cells_to_edges(cells.to(device), deform)
            self._cells_key, self._cells_edges = cells, (s.contiguous(), r.contiguous())
        return self._cells_edges

    @staticmethod
    def _refresh_due(step: int, num_steps: int, frequency) -> bool:
        """A stage configured with `frequency` f recomputes its once-per-period state (balance edges / clusters) at the steps
        that are multiples of ceil(num_steps / f): f = 1 -> only at step 0 of a trajectory."""
        return step % math.ceil(num_steps / frequency) == 0

    def expand_graph(self, graph: MultiGraphWithPos, step: int, num_steps: int, is_training: bool) -> MultiGraph:
        """The optional stages between build_graph and the network, in the reference's order (flag.py:130-141,
        cylinder.py:108-119, plate.py:202-216): graph balancer first (appends the `balance` edge set, renormalises the mesh
        edges), then remote message passing (hyper nodes + remote edge sets)."""
        if self._balancer:
            if self._refresh_due(step, num_steps, self._balance_frequency):
                self._graph_balancer.reset_balancer()
            graph = self._graph_balancer.create_graph(graph, self._mesh_edge_normalizer, is_training)
        if self._rmp:
            if self._refresh_due(step, num_steps, self._rmp_frequency):
                self._remote_graph.reset_clusters()
            graph = self._remote_graph.create_graph(graph, is_training)
        return graph

    def forward(self, graph):
        # Rollout / evaluation (no gradients, on the GPU): the network is replayed from a HIP graph captured per topology, from the
        # second time a topology is seen (graphs.GraphedForwardCache: bit-identical to the eager launches, half their time at one
        # graph per step).  `model.replay_rollout = False` keeps every launch eager.
        if (self.replay_rollout and not torch.is_grad_enabled() and graph.node_features[0].is_cuda
                and not torch.cuda.is_current_stream_capturing()):
            if self._fwd_cache is None:
                from . import graphs
                self._fwd_cache = graphs.GraphedForwardCache(self.learned_model)
            return self._fwd_cache(graph)
        return self.learned_model(graph)

    def evaluate(self) -> None:
        """abstract_system_model.py:187-190."""
        for module in (self, self.learned_model):
            module.eval()

    # ---- evaluation helpers shared by the three models ------------------------------------------------------------
    @staticmethod
    def _first_frame(trajectory: Dict[str, Tensor]) -> Dict[str, Tensor]:
        """Frame 0 of every series of a (possibly batch-of-one) trajectory, on the device."""
        return {name: torch.squeeze(series, 0)[0].to(device) for name, series in trajectory.items()}

    @staticmethod
    def _per_step_mse(truth: Tensor, predicted: Tensor) -> Tensor:
        """[T, N, D] x 2 -> [T]: squared error averaged over components, then over nodes (the reference's two nested means)."""
        return ((truth - predicted) ** 2).mean(dim=-1).mean(dim=-1).detach()

    @torch.no_grad()
    def n_step_computation(self, trajectory: Dict[str, Tensor], n_step: int, num_timesteps=None) -> Tuple[Tensor, Tensor]:
        """Sliding n-step rollouts (flag.py:248-260): every window of n_step + 1 consecutive frames is rolled out from its
        first frame; returns (mean over windows of the window's mean error, mean over windows of its final-step error)."""
        horizon = n_step + 1
        frames = trajectory['cells'].shape[0] if num_timesteps is None else num_timesteps
        window_means, window_finals = [], []
        for start in range(frames - n_step):
            window = {name: series[start:start + horizon] for name, series in trajectory.items()}
            errors = self.rollout(window, horizon)[1].cpu()
            window_means.append(errors.mean())
            window_finals.append(errors[-1])
        return torch.stack(window_means).mean(), torch.stack(window_finals).mean()


class FlagModel(AbstractSystemModel):
    """src/model/flag.py:17-260."""
    _model_type = 'flag'
    _TYPE_MAP = (0,) + (1,) * 9                  # flag.py:72: class = (node_type != NORMAL)

    def __init__(self, params):
        super().__init__(params, node_size=5, edge_size=7)

    def build_graph(self, inputs: Dict, is_training: bool) -> MultiGraphWithPos:
        """flag.py:65-128."""
        world_pos = inputs['world_pos'].to(device)
        prev_world_pos = inputs['prev|world_pos'].to(device)
        mesh_pos = inputs['mesh_pos'].to(device)
        node_type = inputs['node_type'].to(device)
        num_nodes = node_type.shape[0]
        # velocity (3) | one-hot(type != NORMAL) (2)                                              flag.py:68-74
        node_features = features.node_features(world_pos, prev_world_pos, node_type, self._TYPE_MAP, 2)
        senders, receivers = self._mesh_edges(inputs['cells'])
        edge_features, length = features.rel_edge_features(world_pos, mesh_pos, senders, receivers, want_len=True)
        mesh_edges = EdgeSet(name='mesh_edges', features=self._mesh_edge_normalizer(edge_features, is_training),
                             receivers=receivers, senders=senders)
        # max - min incident edge length per node: both aggregates in one pass                      flag.py:100-115
        csr = topology.segment_csr(receivers, num_nodes, world_pos.device)
        mm = ops.aggregate([length.unsqueeze(1)], [(csr.perm, csr.rowptr, csr.seg)], ('max', 'min'))
        node_dynamic = self._node_dynamic_normalizer(features.lincomb3(mm[:, 0], 1.0, mm[:, 1], -1.0))
        return MultiGraphWithPos(
            node_features=[self._node_normalizer(node_features, is_training)], edge_sets=[mesh_edges],
            target_feature=world_pos, mesh_features=mesh_pos, model_type=self._model_type, node_dynamic=node_dynamic,
            unnormalized_edges=EdgeSet(name='mesh_edges', features=edge_features, receivers=receivers, senders=senders),
            obstacle_nodes=None)

    def build_graph_batch(self, inputs: Dict, is_training: bool) -> MultiGraphWithPos:
        """Not in the reference (which builds one graph per frame in Python and concatenates them with
        MeshSimulator._get_batched): B frames of ONE mesh -- `world_pos`, `prev|world_pos`, `node_type` with a leading batch
        dimension [B, N, .], `mesh_pos` [N, 2] or [B, N, 2], `cells` [F, 3] -- become the disjoint union of B graphs in the same
        handful of launches a single frame takes (node ids of frame b are offset by b*N, as batching.batch_graphs does).
        Semantic difference to B separate build_graph calls: each normaliser accumulates ONCE, with the statistics of the
        whole batch (same running sums and counts afterwards, `num_accumulations` grows by 1 instead of B)."""
        world_pos = inputs['world_pos'].to(device)
        B, N = world_pos.shape[0], world_pos.shape[1]
        prev = inputs['prev|world_pos'].to(device).reshape(B * N, 3)
        mesh_pos = inputs['mesh_pos'].to(device)
        mesh_pos = (mesh_pos if mesh_pos.dim() == 3 else mesh_pos.unsqueeze(0).expand(B, N, -1)).reshape(B * N, -1).contiguous()
        node_type = inputs['node_type'].to(device).reshape(B * N, -1)
        world_pos = world_pos.reshape(B * N, 3)
        s1, r1 = self._mesh_edges(inputs['cells'])
        key = (B, N, s1.data_ptr())
        if getattr(self, '_batch_edges_key', None) != key:               # batched topology: once per (mesh, batch size)
            off = (torch.arange(B, device=s1.device) * N).repeat_interleave(s1.shape[0])
            self._batch_edges = ((s1.repeat(B) + off).contiguous(), (r1.repeat(B) + off).contiguous())
            self._batch_edges_key = key
        senders, receivers = self._batch_edges
        node_features = features.node_features(world_pos, prev, node_type, self._TYPE_MAP, 2)
        edge_features, length = features.rel_edge_features(world_pos, mesh_pos, senders, receivers, want_len=True)
        mesh_edges = EdgeSet(name='mesh_edges', features=self._mesh_edge_normalizer(edge_features, is_training),
                             receivers=receivers, senders=senders)
        csr = topology.segment_csr(receivers, B * N, world_pos.device)
        mm = ops.aggregate([length.unsqueeze(1)], [(csr.perm, csr.rowptr, csr.seg)], ('max', 'min'))
        node_dynamic = self._node_dynamic_normalizer(features.lincomb3(mm[:, 0], 1.0, mm[:, 1], -1.0))
        return MultiGraphWithPos(
            node_features=[self._node_normalizer(node_features, is_training)], edge_sets=[mesh_edges],
            target_feature=world_pos, mesh_features=mesh_pos, model_type=self._model_type, node_dynamic=node_dynamic,
            unnormalized_edges=EdgeSet(name='mesh_edges', features=edge_features, receivers=receivers, senders=senders),
            obstacle_nodes=None)

    def _loss_mask(self, data_frame):
        return torch.eq(data_frame['node_type'].to(device)[:, 0], NodeType.NORMAL.value)

    def training_step(self, graph, data_frame):
        """flag.py:146-154."""
        network_output = self(graph)
        target_normalized = self.get_target(data_frame)
        return _masked_mse(target_normalized, network_output, self._loss_mask(data_frame))

    @torch.no_grad()
    def validation_step(self, graph: MultiGraph, data_frame: Dict) -> Tuple[Tensor, Tensor]:
        """flag.py:156-167."""
        prediction = self(graph)
        target_normalized = self.get_target(data_frame, False)
        mask = self._loss_mask(data_frame)
        acc_loss = _masked_mse(target_normalized, prediction, mask).item()
        predicted_position = self.update(data_frame, prediction)
        pos_error = _masked_mse(data_frame['target|world_pos'].to(device), predicted_position, mask).item()
        return acc_loss, pos_error

    def update(self, inputs: Dict, per_node_network_output: Tensor) -> Tensor:
        """flag.py:169-180: next position = 2 cur + acceleration - prev."""
        acceleration = self._output_normalizer.inverse(per_node_network_output)
        return features.lincomb3(inputs['world_pos'].to(device), 2.0, acceleration, 1.0, inputs['prev|world_pos'], -1.0)

    def get_target(self, data_frame, is_training=True):
        """flag.py:182-190."""
        cur = data_frame['world_pos'].to(device)
        prev = data_frame['prev|world_pos'].to(device)
        tgt = data_frame['target|world_pos'].to(device)
        return self._output_normalizer(features.lincomb3(tgt, 1.0, cur, -2.0, prev, 1.0), is_training)

    @torch.no_grad()
    def rollout(self, trajectory: Dict[str, Tensor], num_steps: int) -> Tuple[Dict[str, Tensor], Tensor]:
        """flag.py:192-225."""
        if num_steps is None:
            num_steps = trajectory['cells'].shape[0]
        start = self._first_frame(trajectory)
        free = self._loss_mask(start).unsqueeze(1).expand(-1, 3)        # NORMAL nodes move; HANDLE nodes keep their position
        prev_pos, cur_pos, visited = start['prev|world_pos'], start['world_pos'], []
        for step in range(num_steps):
            prev_pos, cur_pos, visited = self._step_fn(start, prev_pos, cur_pos, visited, free, step)
        self._visualized = False
        predictions = torch.stack(visited)
        errors = self._per_step_mse(trajectory['world_pos'][:num_steps].to(device), predictions)
        return {'faces': trajectory['cells'], 'mesh_pos': trajectory['mesh_pos'], 'gt_pos': trajectory['world_pos'],
                'pred_pos': predictions}, errors

    @torch.no_grad()
    def _step_fn(self, initial_state, prev_pos, cur_pos, trajectory, mask, step):
        """flag.py:227-246."""
        frame = dict(initial_state)
        frame.update({'prev|world_pos': prev_pos, 'world_pos': cur_pos})
        graph = self.expand_graph(self.build_graph(frame, is_training=False), step, 399, is_training=False)   # 399: flag.py:236
        integrated = self.update(frame, self(graph))
        trajectory.append(cur_pos)                              # the trajectory records the state BEFORE the step
        return cur_pos, torch.where(mask, integrated, cur_pos), trajectory


class CylinderModel(AbstractSystemModel):
    """src/model/cylinder.py:17-240 (its remote edges use the 'plate' feature rule, cylinder.py:34)."""
    _model_type = 'plate'
    _TYPE_MAP = (0, -1, -1, -1, 1, 2, 3)        # cylinder.py:71-74: INFLOW->1, OUTFLOW->2, WALL_BOUNDARY->3

    def __init__(self, params):
        super().__init__(params, node_size=6, edge_size=3)

    def build_graph(self, inputs: Dict, is_training: bool) -> MultiGraphWithPos:
        """cylinder.py:65-106."""
        velocity = inputs['velocity'].to(device)
        mesh_pos = inputs['mesh_pos'].to(device)
        node_type = inputs['node_type'].to(device)
        node_features = features.node_features(velocity, None, node_type, self._TYPE_MAP, 4)
        senders, receivers = self._mesh_edges(inputs['cells'])
        edge_features, _ = features.rel_edge_features(mesh_pos, None, senders, receivers)
        mesh_edges = EdgeSet(name='mesh_edges', features=self._mesh_edge_normalizer(edge_features, is_training),
                             receivers=receivers, senders=senders)
        return MultiGraphWithPos(
            node_features=[self._node_normalizer(node_features, is_training)], edge_sets=[mesh_edges],
            mesh_features=mesh_pos, target_feature=velocity, model_type=self._model_type,
            unnormalized_edges=EdgeSet(name='mesh_edges', features=edge_features, receivers=receivers, senders=senders),
            node_dynamic=[], obstacle_nodes=None)

    def _loss_mask(self, data_frame):
        t = data_frame['node_type'].to(device)[:, 0]
        return torch.logical_or(torch.eq(t, NodeType.OUTFLOW.value), torch.eq(t, NodeType.NORMAL.value))

    def training_step(self, graph, data_frame):
        """cylinder.py:123-136."""
        network_output = self(graph)
        target_normalized = self.get_target(data_frame)
        return _masked_mse(target_normalized, network_output, self._loss_mask(data_frame))

    @torch.no_grad()
    def validation_step(self, graph: MultiGraph, data_frame: Dict) -> Tuple[Tensor, Tensor]:
        """cylinder.py:138-153."""
        prediction = self(graph)
        target_normalized = self.get_target(data_frame, False)
        mask = self._loss_mask(data_frame)
        vel_loss = _masked_mse(target_normalized, prediction, mask).item()
        velocity_update, _ = self.update(data_frame, prediction)
        pos_error = _masked_mse(data_frame['target|velocity'].to(device), velocity_update, mask).item()
        return vel_loss, pos_error

    def update(self, inputs: Dict, per_node_network_output: Tensor):
        """cylinder.py:155-165."""
        out = self._output_normalizer.inverse(per_node_network_output)
        velocity, pressure = out[:, :2], out[:, 2:]
        return features.lincomb3(inputs['velocity'].to(device), 1.0, velocity, 1.0), pressure

    def get_target(self, data_frame, is_training=True):
        """cylinder.py:167-173."""
        dv = features.lincomb3(data_frame['target|velocity'].to(device), 1.0, data_frame['velocity'], -1.0)
        return self._output_normalizer(torch.cat((dv, data_frame['pressure'].to(device)), dim=1), is_training)

    @torch.no_grad()
    def rollout(self, trajectory: Dict[str, Tensor], num_steps: int):
        """cylinder.py:175-208."""
        num_steps = trajectory['cells'].shape[0]                          # the whole trajectory, whatever was asked (cylinder.py:178)
        start = self._first_frame(trajectory)
        free = self._loss_mask(start).unsqueeze(1).expand(-1, 2)        # NORMAL and OUTFLOW nodes are integrated, the rest is prescribed
        velocity, pressure, velocities, pressures = start['velocity'], start['pressure'], [], []
        for step in range(num_steps):
            velocity, pressure, velocities, pressures = self._step_fn(start, velocity, pressure, velocities, pressures, step, free)
        pred_velocity = torch.stack(velocities)
        errors = self._per_step_mse(trajectory['velocity'][:num_steps].to(device), pred_velocity)
        return {'faces': trajectory['cells'], 'mesh_pos': trajectory['mesh_pos'], 'gt_velocity': trajectory['velocity'],
                'gt_pressure': trajectory['pressure'], 'pred_pressure': torch.stack(pressures), 'pred_velocity': pred_velocity}, errors

    @torch.no_grad()
    def _step_fn(self, initial_state, velocity, pressure, trajectory, pressure_trajectory, step, mask):
        """cylinder.py:210-230."""
        frame = dict(initial_state)
        frame.update({'velocity': velocity, 'pressure': pressure})
        graph = self.expand_graph(self.build_graph(frame, is_training=False), step, 598, is_training=False)   # 598: cylinder.py:218
        integrated, new_pressure = self.update(frame, self(graph))
        new_velocity = torch.where(mask, integrated, velocity)
        trajectory.append(new_velocity)                         # here the state AFTER the step is recorded
        pressure_trajectory.append(new_pressure)
        return new_velocity, new_pressure, trajectory, pressure_trajectory


class PlateModel(AbstractSystemModel):
    """src/model/plate.py:17-340 (deforming plate: world edges from obstacle to plate nodes, 4-vertex cells)."""
    _model_type = 'plate'
    _TYPE_MAP = (0, 1, 2, 2)                     # plate.py:78: HANDLE (3) -> class 2
    _RADIUS = 0.03                               # plate.py:85

    def __init__(self, params):
        super().__init__(params, node_size=6, edge_size=8, remote_edge_size=8, edge_sets=('mesh_edges', 'world_edges'))
        self._world_edge_normalizer = Normalizer(size=4, name='world_edge_normalizer')

    def _select_architecture(self, connector):
        return connector if (self._rmp or connector == 'repeated') else 'none'        # plate.py:37-39

    def build_graph(self, inputs: Dict, is_training: bool) -> MultiGraphWithPos:
        """plate.py:69-200."""
        world_pos = inputs['world_pos'].to(device)
        mesh_pos = inputs['mesh_pos'].to(device)
        target_world_pos = inputs['target|world_pos'].to(device)
        node_type = inputs['node_type'].to(device)
        num_nodes = node_type.shape[0]
        senders, receivers = self._mesh_edges(inputs['cells'], deform=True)
        # world edges: obstacle -> normal pairs closer than the radius that are not mesh edges          plate.py:84-110
        csr = topology.segment_csr(receivers, num_nodes, world_pos.device)        # neighbours of n = senders of its edges
        if getattr(self, '_nbr_key', None) is not senders:
            self._nbr = senders[csr.perm.long()].to(torch.int32).contiguous()
            self._nbr_key = senders
        world_senders, world_receivers = features.radius_edges(world_pos, node_type, self._RADIUS, NodeType.OBSTACLE.value,
                                                                NodeType.NORMAL.value, csr.rowptr, self._nbr)
        world_edge_features, _ = features.rel_edge_features(world_pos, None, world_senders, world_receivers)
        world_edges = EdgeSet(name='world_edges', features=self._world_edge_normalizer(world_edge_features, is_training),
                              receivers=world_receivers, senders=world_senders)
        mesh_edge_features, _ = features.rel_edge_features(world_pos, mesh_pos, senders, receivers)
        mesh_edges = EdgeSet(name='mesh_edges', features=self._mesh_edge_normalizer(mesh_edge_features, is_training),
                             receivers=receivers, senders=senders)
        # one-hot(3) | velocity of the kinematic (obstacle) nodes, zero elsewhere                      plate.py:186-195
        node_features = features.node_features(target_world_pos, world_pos, node_type, self._TYPE_MAP, 3, vel_first=False,
                                               vel_mask_type=NodeType.OBSTACLE.value)
        obstacle_nodes = torch.eq(node_type[:, 0], NodeType.OBSTACLE.value)
        return MultiGraphWithPos(
            node_features=[self._node_normalizer(node_features, is_training)], edge_sets=[mesh_edges, world_edges],
            mesh_features=mesh_pos, target_feature=world_pos, model_type=self._model_type,
            unnormalized_edges=EdgeSet(name='mesh_edges', features=mesh_edge_features, receivers=receivers,
                                       senders=senders),
            node_dynamic=None, obstacle_nodes=obstacle_nodes)

    def _loss_mask(self, data_frame):
        return torch.eq(data_frame['node_type'].to(device)[:, 0], NodeType.NORMAL.value)

    def training_step(self, graph, data_frame):
        """plate.py:218-228."""
        network_output = self(graph)
        target_normalized = self.get_target(data_frame)
        return _masked_mse(target_normalized, network_output, self._loss_mask(data_frame))

    @torch.no_grad()
    def validation_step(self, graph: MultiGraph, data_frame: Dict) -> Tuple[Tensor, Tensor]:
        """plate.py:230-244."""
        prediction = self(graph)
        target_normalized = self.get_target(data_frame, False)
        mask = self._loss_mask(data_frame)
        vel_loss = _masked_mse(target_normalized, prediction, mask).item()
        predicted_position, _, _ = self.update(data_frame, prediction)
        pos_error = _masked_mse(data_frame['target|world_pos'].to(device), predicted_position, mask).item()
        return vel_loss, pos_error

    def update(self, inputs: Dict, per_node_network_output: Tensor):
        """plate.py:246-257: next position = current + predicted velocity."""
        velocity = self._output_normalizer.inverse(per_node_network_output)
        cur_position = inputs['world_pos'].to(device)
        return features.lincomb3(cur_position, 1.0, velocity, 1.0), cur_position, velocity

    def get_target(self, data_frame, is_training=True):
        """plate.py:259-264."""
        v = features.lincomb3(data_frame['target|world_pos'].to(device), 1.0, data_frame['world_pos'], -1.0)
        return self._output_normalizer(v, is_training)

    @torch.no_grad()
    def rollout(self, trajectory: Dict[str, Tensor], num_steps: int):
        """plate.py:266-316."""
        if num_steps is None:
            num_steps = trajectory['cells'].shape[0]
        start = self._first_frame(trajectory)
        free = self._loss_mask(start).unsqueeze(1).expand(-1, 3)        # NORMAL nodes are predicted; obstacle / handle nodes are scripted
        scripted = trajectory['target|world_pos'].to(device)
        position, predicted, positions, velocities = start['world_pos'], [], [], []
        for step in range(num_steps):
            position, predicted, positions, velocities = self._step_fn(start, position, predicted, positions, velocities,
                                                                       scripted[step], step, free, num_steps)
        pred_pos = torch.stack(predicted)
        errors = self._per_step_mse(trajectory['world_pos'][:num_steps].to(device), pred_pos)
        # the viewer wants triangles: every 4-vertex cell (v0 v1 v2 v3) contributes (v0 v1 v2) and (v2 v3 v0)   plate.py:289-297
        cells = trajectory['cells']
        triangles = torch.cat((cells[..., 0:3], cells[..., [2, 3, 0]]), dim=-2)
        return {'faces': triangles, 'mesh_pos': trajectory['mesh_pos'],
                'mask': torch.eq(start['node_type'][:, 0], NodeType.OBSTACLE.value), 'gt_pos': trajectory['world_pos'],
                'pred_pos': pred_pos, 'cur_positions': torch.stack(positions), 'cur_velocities': torch.stack(velocities)}, errors

    @torch.no_grad()
    def _step_fn(self, initial_state, cur_pos, trajectory, cur_positions, cur_velocities, target_world_pos, step, mask,
                 num_steps):
        """plate.py:318-340."""
        frame = dict(initial_state)
        frame.update({'world_pos': cur_pos, 'target|world_pos': target_world_pos})
        graph = self.expand_graph(self.build_graph(frame, is_training=False), step, num_steps, is_training=False)
        integrated, position_before, velocity = self.update(frame, self(graph))
        new_pos = torch.where(mask, integrated, target_world_pos)       # scripted nodes follow the prescribed motion
        trajectory.append(new_pos)
        cur_positions.append(position_before)
        cur_velocities.append(velocity)
        return new_pos, trajectory, cur_positions, cur_velocities


def get_model(config) -> AbstractSystemModel:
    """src/model/get_model.py:13-22."""
    name = str(config['task']['dataset']).lower()
    if 'flag' in name:
        return FlagModel(config.get('model'))
    if 'plate' in name:
        return PlateModel(config.get('model'))
    if 'cylinder' in name:
        return CylinderModel(config.get('model'))
    raise NotImplementedError('Implement your algorithms here!')
