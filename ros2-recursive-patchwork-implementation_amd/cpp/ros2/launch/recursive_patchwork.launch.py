#!/usr/bin/env python3
"""Launch file of recursive_patchwork_node (MI355X build): the launch arguments of the reference's
launch/recursive_patchwork.launch.py:11-50 with the same names and defaults, plus the two that select the GPU and the
number of frames in flight."""

from launch import LaunchDescription
from launch.actions import DeclareLaunchArgument
from launch.substitutions import LaunchConfiguration
from launch_ros.actions import Node

ARGUMENTS = [
    # name, default, description                                  (reference launch file, same order)
    ('input_topic', '/lidar/points', 'Input point cloud topic'),
    ('ground_topic', '/patchwork/ground', 'Output ground points topic'),
    ('obstacles_topic', '/patchwork/obstacles', 'Output obstacle points topic'),
    ('visualization_topic', '/patchwork/visualization', 'Output visualization markers topic'),
    ('min_points', '100', 'Minimum points for plane fitting'),
    ('max_iterations', '50', 'Maximum iterations for plane fitting'),
    ('distance_threshold', '0.1', 'Distance threshold for plane fitting'),
    ('angle_threshold', '0.1', 'Angle threshold for plane fitting'),
    # additions of this build
    ('device', '0', 'HIP device ordinal of the MI355X to use'),
    ('frames_in_flight', '1', 'Frames processed concurrently (1 = every callback publishes its own frame, as the '
                              'reference does; k > 1 = results are published k - 1 callbacks later, in order)'),
]


def generate_launch_description():
    declared = [DeclareLaunchArgument(name, default_value=default, description=text) for name, default, text in ARGUMENTS]
    node = Node(
        package='recursive_patchwork',
        executable='recursive_patchwork_node',
        name='recursive_patchwork_node',
        output='screen',
        parameters=[{name: LaunchConfiguration(name) for name, _, _ in ARGUMENTS}],
        remappings=[
            ('/lidar/points', LaunchConfiguration('input_topic')),
            ('/patchwork/ground', LaunchConfiguration('ground_topic')),
            ('/patchwork/obstacles', LaunchConfiguration('obstacles_topic')),
            ('/patchwork/visualization', LaunchConfiguration('visualization_topic')),
        ],
    )
    return LaunchDescription(declared + [node])
